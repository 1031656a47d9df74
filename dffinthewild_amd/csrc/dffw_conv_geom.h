// Device-side geometry/tile templates of conv_tile.
#pragma once
#include "dffw_conv_tile.h"
#include "dffw_device.h"

namespace dffw {

template <int GEO>
struct GeoT;
template <>
struct GeoT<G3S1> { static constexpr int MINZ = -1, MAXZ = 1, MINY = -1, MAXY = 1, S = 1, OS = 1, NPASS = 1, MINX = MINY, MAXX = MAXY; };
template <>
struct GeoT<G3S2> { static constexpr int MINZ = -1, MAXZ = 1, MINY = -1, MAXY = 1, S = 2, OS = 1, NPASS = 1, MINX = MINY, MAXX = MAXY; };
template <>
struct GeoT<G3T> { static constexpr int MINZ = -1, MAXZ = 1, MINY = 0, MAXY = 1, S = 1, OS = 2, NPASS = 4, MINX = MINY, MAXX = MAXY; };
template <>
struct GeoT<G2S1> { static constexpr int MINZ = 0, MAXZ = 0, MINY = -1, MAXY = 1, S = 1, OS = 1, NPASS = 1, MINX = MINY, MAXX = MAXY; };
template <>
struct GeoT<G2S2> { static constexpr int MINZ = 0, MAXZ = 0, MINY = -1, MAXY = 1, S = 2, OS = 1, NPASS = 1, MINX = MINY, MAXX = MAXY; };
template <>
struct GeoT<G2D> { static constexpr int MINZ = 0, MAXZ = 0, MINY = -8, MAXY = 8, S = 1, OS = 1, NPASS = 1, MINX = -6, MAXX = 10; };

template <>
struct GeoT<G2P> { static constexpr int MINZ = 0, MAXZ = 0, MINY = -8, MAXY = 8, S = 1, OS = 1, NPASS = 1, MINX = -6, MAXX = 10; };

template <int GEO, int TZ_, int TY_, int TX_, int CG_>
struct TileT {
    using G = GeoT<GEO>;
    static constexpr int TZ = TZ_, TY = TY_, TX = TX_, CG = CG_;
    static constexpr int FZ = TZ + G::MAXZ - G::MINZ;
    static constexpr int FY = (TY - 1) * G::S + (G::MAXY - G::MINY) + 1;
    static constexpr int FX = (TX - 1) * G::S + (G::MAXX - G::MINX) + 1;
    // pixel-pair stem: an operand tile is 16 PAIRS (x, x+2) with x = 0,1 mod 4, and only the records at columns = 0,1 mod 4 of
    // the footprint are ever read (x + 2*jx for the five even filter columns jx): the LDS image keeps those alone, packed
    static constexpr bool PAIR = (GEO == G2P);
    static constexpr int FXL = PAIR ? FX / 2 : ((G::S == 2) ? (FX + 1) / 2 * 2 : FX);
    static constexpr int FPIX = FZ * FY * FXL;
    static constexpr int MT = TZ * TY * TX / (PAIR ? 32 : 16);
    static_assert(TZ * TY * TX % (PAIR ? 128 : 64) == 0, "tile must split evenly over 4 waves of 16-point operand tiles");
    static_assert(!PAIR || (TX % 8 == 0 && FX % 4 == 0), "pair form: whole groups of 8 pixels per row");
};

}  // namespace dffw
