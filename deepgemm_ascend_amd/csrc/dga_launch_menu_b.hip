// fp8 tile-kernel menu, part B: the continuous-pipeline builds of the smaller tiles and the loader / consumer build (dga_fp8_menu.hpp).
#include "dga_fp8_menu_impl.hpp"
namespace dga {
DGA_MENU_B(DGA_MENU_INSTANTIATE)
DGA_MENU_LC(DGA_MENU_INSTANTIATE_LC)
}
