// fp8 tile-kernel menu, part B: the continuous-pipeline builds of the smaller tiles (dga_fp8_menu.hpp).
#include "dga_fp8_menu_impl.hpp"
namespace dga {
DGA_MENU_B(DGA_MENU_INSTANTIATE)
}
