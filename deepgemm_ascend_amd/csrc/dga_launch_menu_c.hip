// fp8 tile-kernel menu, part C: the one-barrier-per-k-block builds (2 and 3 LDS stages) of every tile below 256x256.
#include "dga_fp8_menu_impl.hpp"
namespace dga {
DGA_MENU_C(DGA_MENU_INSTANTIATE)
}
