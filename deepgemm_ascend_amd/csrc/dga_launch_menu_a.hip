// fp8 tile-kernel menu, part A: the 256x256 tile in its three main-loop schedules (dga_fp8_menu.hpp).
#include "dga_fp8_menu_impl.hpp"
namespace dga {
DGA_MENU_A(DGA_MENU_INSTANTIATE)
}
