// conv_wino32 (Winograd F(2x2, 3x3) on the in-plane taps of a 3x3x3 stride-1 conv over 32 input channels, dffw_conv_wino.hip):
// host/device declarations.
#pragma once
#include "dffw_internal.h"

namespace dffw {

struct WinoArgs {
    const uint16_t *u;       // transformed filter U = G g G^T in MFMA fragment order [slab of 32 outputs][position 16][dz 3][nt 2][part][64 lanes][8]
    int tiles_y, tiles_x;    // 4 x 16 output-pixel columns per sample
};

constexpr int WINO_TY = 4, WINO_TX = 16;   // a column's footprint on the output grid: 2 x 8 blocks of 2 x 2 pixels
// elements of WinoArgs::u per 32-output slab
constexpr size_t WINO_U_SLAB = (size_t)16 * 3 * 2 * 2 * 512;

// grid: x = B * tiles_y * tiles_x columns, y = Cout / 32 slabs
hipError_t launch_conv_wino32(int prec, const ConvArgs &a, const WinoArgs &t, hipStream_t s);
void conv_wino32_kernel_name(int prec, const ConvArgs &a, char *buf, int n);

}  // namespace dffw
