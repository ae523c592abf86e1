// Weight-gradient kernels for gfx950 (MI355X): split-K over pixels with LDS transpose reads, grouped / multi / persistent
// launches and their launch planner.  (Split out of conv.hip in round 5; the forward / data-gradient kernels stay there.)
//
// Replaces the autograd weight / bias gradients of F.conv2d at mmdet/models/backbones/resnet.py:262-301,598-645,
// mmdet/models/necks/fpn.py:150-202, mmdet/models/dense_heads/anchor_free_head.py:197-217 and fcos_head.py:154-156.
//
// Layout: activations and gradients NHWC bf16; dW fp32 [Cout][kh][kw][Cin] written straight into the flat gradient buffer.
#include <stdlib.h>
#include <algorithm>

#include <mutex>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(16))) unsigned int g_zero_line[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ================================================================================================
// weight gradient
// ================================================================================================
struct WgK {
  int nseg, n;
  int gh[DSL_MAX_SEG], gw[DSL_MAX_SEG], sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];
  int pxstart[DSL_MAX_SEG + 1];
  long long xoff[DSL_MAX_SEG];
  FastDiv dhw[DSL_MAX_SEG], dwd[DSL_MAX_SEG];
  int cs, cy, kh, kw, stride, pad;
  int ktiles, tiles_per_split, ctiles_per_tap;
  int dbg;                  // ablation knobs (ablation build only): 1 = skip DMA after the first tile, 2 = skip MFMA, 4 = no epilogue
  int gx, gy, splits;       // v2: workgroup grid (cout tiles, column tiles) and split count for the XCD-aware 1-D launch
  int chunk;                // v2: consecutive work items (split-major) per XCD
  int group;                // v2: convolutions sharing this geometry in one launch (dsl_conv2d_wgrad_group)
  int ldx;                  // pixel stride of X in elements (>= cs)
  int cyp;                  // v2: cy rounded up to the cout tile (partial-tile rows in the workspace); dY columns >= cy read as zero
  const uint16_t* dyv[DSL_MAX_GROUP];
  const uint16_t* xv[DSL_MAX_GROUP];
  long long krow;
  const uint16_t* dy;
  const uint16_t* x;
  float* ws;
  // multi-launch (dsl_conv2d_wgrad_multi): with direct = 1 (splits == 1) the finished tile goes straight into dW (x scale)
  int direct, cd;
  int totpx;                // = pxstart[nseg] (a runtime-indexed read would keep a table copy of this struct in scratch)
  float* dwv[DSL_MAX_GROUP];
  const float* scalev[DSL_MAX_GROUP];
  // bias gradients db[co] = sum over pixels of dY[.][co], summed by the tap-0 / first-cin-tile workgroups from the dY stages
  // they stream anyway: dbmask bit g = member g has a db; partial sums go to dbws[(split * group + member) * cyp + co]
  // (direct launches: straight into dbv[member]).  Fixed summation order, no atomics.
  int dbmask;
  float* dbws;
  float* dbv[DSL_MAX_GROUP];
  // v3 (wgrad_pipe): per-pixel gather descriptors of this geometry (PixDesc, one per dY pixel, built once per geometry)
  const void* pixtab;
  unsigned pixtab_bytes;
  unsigned ybytes;          // extent of one member's dY in bytes (= totpx * cy * 2): rows past it read as zeros
};

template <int ROWBYTES>
__device__ __forceinline__ int tr_swz(int row) {
  return ROWBYTES == 256 ? (row & 3) : ((row >> 1) & 1);
}

// BCO couts x 128 cins per workgroup, 64 pixels per K stage; both operands are stored
// [pixel][channel] in LDS and read with ds_read_b64_tr_b16 (hardware transpose) into MFMA fragments.
template <int BCO>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int YB = BCO * 2;             // bytes per pixel row of the dY tile
  constexpr int XB = 256;                 // 128 cin * 2
  constexpr int TILE_Y = 64 * YB, TILE_X = 64 * XB, STAGE = TILE_Y + TILE_X;
  constexpr int WM = BCO / 2, CT = WM / 32;
  constexpr int YCPR = BCO / 8;           // 16-byte chunks per dY row
  constexpr int YRPP = 256 / YCPR;        // rows per pass
  constexpr int YPASS = 64 / YRPP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_co = wave >> 1, wave_ci = wave & 1;
  const int co0 = blockIdx.x * BCO;
  const int colt = blockIdx.y;
  const int tap = colt / p.ctiles_per_tap;
  const int ci0 = (colt - tap * p.ctiles_per_tap) * 128;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int sp = blockIdx.z;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  const int totpx = p.pxstart[p.nseg];

  const int yrow = tid / YCPR, ychunk = tid % YCPR;
  const int xrow = tid >> 4, xchunk = tid & 15;
  u32x4 ry[YPASS], rx[4];

  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < YPASS; ++i) {
      const int gp = kt * 64 + yrow + YRPP * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gp < totpx) v = *reinterpret_cast<const u32x4*>(p.dy + (long long)gp * p.cy + co0 + ychunk * 8);
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gp = kt * 64 + xrow + 16 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gp < totpx) {
        int seg = 0;
#pragma unroll
        for (int s = 1; s < DSL_MAX_SEG; ++s)
          if (s < p.nseg && gp >= p.pxstart[s]) seg = s;
        const uint32_t q = gp - p.pxstart[seg];
        const uint32_t img = fdiv(q, p.dhw[seg]);
        const uint32_t rem = q - img * p.dhw[seg].d;
        const uint32_t y = fdiv(rem, p.dwd[seg]);
        const uint32_t x = rem - y * p.dwd[seg].d;
        const int sy = (int)y * p.stride + tr - p.pad, sx = (int)x * p.stride + ts - p.pad;
        if ((unsigned)sy < (unsigned)p.sh[seg] && (unsigned)sx < (unsigned)p.sw[seg]) {
          const long long pix = p.xoff[seg] + ((long long)img * p.sh[seg] + sy) * p.sw[seg] + sx;
          v = *reinterpret_cast<const u32x4*>(p.x + pix * p.ldx + ci0 + xchunk * 8);
        }
      }
      rx[i] = v;
    }
  };
  auto lds_store = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < YPASS; ++i) {
      const int row = yrow + YRPP * i;
      *reinterpret_cast<u32x4*>(base + row * YB + ((((ychunk >> 2) ^ tr_swz<YB>(row))) << 6) + ((ychunk & 3) << 4)) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = xrow + 16 * i;
      *reinterpret_cast<u32x4*>(base + TILE_Y + row * XB + ((((xchunk >> 2) ^ tr_swz<XB>(row))) << 6) + ((xchunk & 3) << 4)) = rx[i];
    }
  };

  f32x16 acc[CT][2];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  // transpose-read geometry: 16-lane group g covers channel block (g&1)*16 and pixel block (g>>1)*8
  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);           // + kk*16 + hh*4
  const int ccol_l = iblk + 4 * (l16 & 3);        // channel (element) offset inside a 32-wide tile

  auto tr_read = [&](const unsigned char* tile, int rowbytes_sel, int krow, int col) -> s16x4 {
    int byte;
    if (rowbytes_sel == 256)
      byte = krow * 256 + ((((col * 2) >> 6) ^ tr_swz<256>(krow)) << 6) + ((col * 2) & 63);
    else
      byte = krow * 128 + ((((col * 2) >> 6) ^ tr_swz<128>(krow)) << 6) + ((col * 2) & 63);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + byte));
  };

  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 a[CT], b[2];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = wave_co * WM + ct * 32 + ccol_l;
        const s16x4 lo = tr_read(base, YB, kk * 16 + krow_l, col);
        const s16x4 hi = tr_read(base, YB, kk * 16 + krow_l + 4, col);
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo;
        u.s.h = hi;
        a[ct] = u.v;
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int col = wave_ci * 64 + pt * 32 + ccol_l;
        const s16x4 lo = tr_read(base + TILE_Y, XB, kk * 16 + krow_l, col);
        const s16x4 hi = tr_read(base + TILE_Y, XB, kk * 16 + krow_l + 4, col);
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo;
        u.s.h = hi;
        b[pt] = u.v;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
    }
  };

  if (kt0 < kt1) {
    gload(kt0);
    lds_store(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      if (kt + 1 < kt1) gload(kt + 1);
      compute(cur);
      if (kt + 1 < kt1) lds_store(cur ^ 1);
      __syncthreads();
    }
  }

  // partial tile -> workspace [split][cy][krow]
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * 64 + pt * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * WM + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[((long long)sp * p.cy + co) * p.krow + col] = acc[ct][pt][j];
      }
    }
}

// inline-asm helpers must be explicit __device__ functions: a lambda inside a kernel is implicitly
// __host__ __device__, and its AMDGPU asm constraints break the (silently dropped) host instantiation
__device__ __forceinline__ u32x2 lds_tr_read_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// ... with the constant part of the address in the instruction's 16-bit offset field (inline asm is opaque to the compiler: given
// the whole address in a register it spends one v_add per read - 24 of the 33 VALU instructions per stage of wgrad_pipe's K loop)
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_read_b64_o(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "DS offset field");
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// read (k-step kk, half hi) of a fragment column whose (stage, column) address is `base`: kk and hi are constants after unrolling
template <int ROWB, int KK>
__device__ __forceinline__ u32x2 lds_tr_read_kh(unsigned base, int kk, int hi) {
  static_assert(KK <= 4, "k-steps per stage");
  switch (kk * 2 + hi) {
    case 0: return lds_tr_read_b64_o<0>(base);
    case 1: return lds_tr_read_b64_o<4 * ROWB>(base);
    case 2: return lds_tr_read_b64_o<16 * ROWB>(base);
    case 3: return lds_tr_read_b64_o<20 * ROWB>(base);
    case 4: return lds_tr_read_b64_o<32 * ROWB>(base);
    case 5: return lds_tr_read_b64_o<36 * ROWB>(base);
    case 6: return lds_tr_read_b64_o<48 * ROWB>(base);
    default: return lds_tr_read_b64_o<52 * ROWB>(base);
  }
}
struct Frag {
  u32x2 lo, hi;
};
__device__ __forceinline__ unsigned lds_read_b32_asm(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void wait_lds4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
// Wait for the outstanding transpose reads AND tell the compiler the fragment registers change here
// ("+v"): otherwise it may copy an asm-loaded register before the data has landed (the destination of an
// inline-asm load counts as written when the statement ends, not when the LDS returns).
template <int CT, int IT>
__device__ __forceinline__ void wait_frags(Frag (&fa)[CT], Frag (&fb)[IT]) {
  static_assert(IT == 2 && (CT == 2 || CT == 4), "fragment shapes used by the wgrad tiles");
  if constexpr (CT == 4) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0].lo), "+v"(fa[0].hi), "+v"(fa[1].lo), "+v"(fa[1].hi), "+v"(fa[2].lo), "+v"(fa[2].hi),
                   "+v"(fa[3].lo), "+v"(fa[3].hi), "+v"(fb[0].lo), "+v"(fb[0].hi), "+v"(fb[1].lo), "+v"(fb[1].hi)
                 :
                 : "memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0].lo), "+v"(fa[0].hi), "+v"(fa[1].lo), "+v"(fa[1].hi), "+v"(fb[0].lo), "+v"(fb[0].hi),
                   "+v"(fb[1].lo), "+v"(fb[1].hi)
                 :
                 : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
}

// v2 weight gradient: DMA-to-LDS operands, 256-wide tiles, 8 waves, NST-deep ring of KS-pixel stages with
// counted vmcnt waits (the pixel streams come from HBM: one stage of lookahead does not cover the latency).
// Same math/outputs as wgrad_kernel.
struct SegSel {          // per-segment decode constants, selected with v_cndmask chains (no memory access:
  int px0;               // indexing kernel-argument arrays or LDS tables by a runtime segment id makes hipcc
  uint32_t m1lo, m1hi, d1, m2lo, m2hi, d2;   // drain the DMA queue with s_waitcnt vmcnt(0) inside the K loop)
  int sh, sw;
  long long xoff;
};
__device__ __forceinline__ SegSel seg_select(const WgK& p, int gp) {
  SegSel r;
  r.px0 = p.pxstart[0];
  r.m1lo = (uint32_t)p.dhw[0].m; r.m1hi = (uint32_t)(p.dhw[0].m >> 32); r.d1 = p.dhw[0].d;
  r.m2lo = (uint32_t)p.dwd[0].m; r.m2hi = (uint32_t)(p.dwd[0].m >> 32); r.d2 = p.dwd[0].d;
  r.sh = p.sh[0]; r.sw = p.sw[0]; r.xoff = p.xoff[0];
#pragma unroll
  for (int s = 1; s < DSL_MAX_SEG; ++s) {
    const bool in = s < p.nseg && gp >= p.pxstart[s];
    r.px0 = in ? p.pxstart[s] : r.px0;
    r.m1lo = in ? (uint32_t)p.dhw[s].m : r.m1lo; r.m1hi = in ? (uint32_t)(p.dhw[s].m >> 32) : r.m1hi;
    r.d1 = in ? p.dhw[s].d : r.d1;
    r.m2lo = in ? (uint32_t)p.dwd[s].m : r.m2lo; r.m2hi = in ? (uint32_t)(p.dwd[s].m >> 32) : r.m2hi;
    r.d2 = in ? p.dwd[s].d : r.d2;
    r.sh = in ? p.sh[s] : r.sh; r.sw = in ? p.sw[s] : r.sw;
    r.xoff = in ? p.xoff[s] : r.xoff;
  }
  return r;
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__device__ __forceinline__ void wgrad_glds_body(const WgK& p, const int bid, unsigned char* smem) {
  constexpr int NW = WCO * WCI;
  constexpr int YB = BCO * 2, XB = BCI * 2;          // bytes per pixel row of each tile
  constexpr int TILE_Y = KS * YB, TILE_X = KS * XB, STAGE = TILE_Y + TILE_X;
  constexpr int NY = TILE_Y / 1024, NX = TILE_X / 1024;      // 1 KB DMA instructions per tile
  constexpr int LY = NY / NW, LX = NX / NW;                   // per wave
  constexpr int LPT = LY + LX;
  constexpr int CT = BCO / WCO / 32, IT = BCI / WCI / 32;
  constexpr int KK = KS / 16;                                  // MFMA k-steps per stage
  static_assert(NY % NW == 0 && NX % NW == 0 && LY >= 1 && LX >= 1, "tile / wave mismatch");
  static_assert(YB >= 256 && XB >= 256, "64-byte-chunk swizzle needs >= 4 chunks per row");
  static_assert(KK == 2 || KK == 4, "stage depth");
  static_assert((NST - 2) * LPT <= 63 && NST >= 2 && NST <= 4, "vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WCI, wave_ci = wave % WCI;
  // XCD-aware work mapping (blocks are dealt round-robin to the XCDs: equal b % 8 = same XCD): the (cout tile, tap, cin tile) workgroups of one pixel
  // split are neighbours on one XCD, so its dY / X pixel range is fetched into that XCD's L2 once instead of
  // once per tap.  Placement only affects speed.
  const int tiles_per_member = p.gx * p.gy;
  const int tiles_per_split_wg = tiles_per_member * p.group;
  const int xcd = bid & 7, jj = bid >> 3;
  const int witem = xcd * p.chunk + jj;              // work items are split-major: an XCD owns a contiguous range
  if (jj >= p.chunk || witem >= tiles_per_split_wg * p.splits) return;
  const int sp = witem / tiles_per_split_wg;
  const int rem_sp = witem - sp * tiles_per_split_wg;
  const int member = rem_sp / tiles_per_member;      // which convolution of the group
  const int rem_wg = rem_sp - member * tiles_per_member;
  // member pointers by select chain, once, outside the K loop (a runtime-indexed kernel-argument load inside the
  // loop would make hipcc drain the DMA queue)
  const uint16_t* dy_p = p.dyv[0];
  const uint16_t* x_p = p.xv[0];
  float* db_p = p.dbv[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dy_p = member == g ? p.dyv[g] : dy_p;
    x_p = member == g ? p.xv[g] : x_p;
    db_p = member == g ? p.dbv[g] : db_p;
  }
  const int co0 = (rem_wg % p.gx) * BCO;
  const int colt = rem_wg / p.gx;
  // column sums of dY (the bias gradient) ride along in the workgroups of column tile 0
  const bool do_db = colt == 0 && ((p.dbmask >> member) & 1);
  constexpr int DB_PAIRS = BCO / 2, DB_RG = 64 * NW / DB_PAIRS, DB_ROWS = KS / DB_RG;
  static_assert(DB_ROWS % 4 == 0 && DB_RG * DB_PAIRS == 64 * NW, "bias-gradient thread mapping");
  const int db_cp = tid % DB_PAIRS, db_rg = tid / DB_PAIRS;
  float db_lo = 0.f, db_hi = 0.f;
  const int ctiles = p.cs / BCI;
  const int tap = colt / ctiles;
  const int ci0 = (colt - tap * ctiles) * BCI;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  const int totpx = p.totpx;
  const gptr_t zero = (gptr_t)g_zero_line;

  // per DMA instruction this lane's (row, source channel) inside the tile
  int yrow[LY], ych[LY], xrow[LX], xch[LX];
#pragma unroll
  for (int i = 0; i < LY; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / YB, inrow = off % YB;
    yrow[i] = row;
    ych[i] = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);      // bf16 element offset in the row
  }
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / XB, inrow = off % XB;
    xrow[i] = row;
    xch[i] = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
  }

  f32x16 acc[CT][IT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < IT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  // Transpose reads are issued through inline asm: given the builtin (an addrspace(3) access) hipcc orders
  // every ds_read behind the in-flight LDS-DMA with s_waitcnt vmcnt(0), which would serialise DMA and MFMA.
  // The hazards are handled by hand: DMA data is read one barrier after its counted vmcnt wait; fragment
  // registers are consumed only after an explicit lgkmcnt(0) naming them.
  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);
  const int ccol_l = iblk + 4 * (l16 & 3);
  const unsigned lds_base = (unsigned)(size_t)smem;      // low 32 bits of a flat LDS address = the LDS offset
  unsigned a_off[CT], b_off[IT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = wave_co * (32 * CT) + ct * 32 + ccol_l;
    a_off[ct] = krow_l * YB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int col = wave_ci * (32 * IT) + it * 32 + ccol_l;
    b_off[it] = TILE_Y + krow_l * XB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
  auto issue = [&](unsigned stage_addr, int kk, Frag (&fa)[CT], Frag (&fb)[IT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const unsigned ad = stage_addr + a_off[ct] + kk * 16 * YB;
      fa[ct].lo = lds_tr_read_b64(ad);
      fa[ct].hi = lds_tr_read_b64_o<4 * YB>(ad);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const unsigned ad = stage_addr + b_off[it] + kk * 16 * XB;
      fb[it].lo = lds_tr_read_b64(ad);
      fb[it].hi = lds_tr_read_b64_o<4 * XB>(ad);
    }
  };
  auto mma = [&](Frag (&fa)[CT], Frag (&fb)[IT]) {
    bf16x8 a[CT], b[IT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fa[ct].lo;
      u.s.h = fa[ct].hi;
      a[ct] = u.v;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fb[it].lo;
      u.s.h = fb[it].hi;
      b[it] = u.v;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it)
        acc[ct][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[it], acc[ct][it], 0, 0, 0);
  };
  auto compute = [&](int slot) {
    const unsigned st = lds_base + slot * STAGE;
    Frag fa0[CT], fb0[IT], fa1[CT], fb1[IT];
    issue(st, 0, fa0, fb0);
    wait_frags<CT, IT>(fa0, fb0);
    issue(st, 1, fa1, fb1);           // next fragments fly while the MFMAs of this step run
    mma(fa0, fb0);
    wait_frags<CT, IT>(fa1, fb1);
    if constexpr (KK == 4) {
      issue(st, 2, fa0, fb0);
      mma(fa1, fb1);
      wait_frags<CT, IT>(fa0, fb0);
      issue(st, 3, fa1, fb1);
      mma(fa0, fb0);
      wait_frags<CT, IT>(fa1, fb1);
    }
    mma(fa1, fb1);
  };

  // DMA of one stage (tile index kt -> ring slot): address math first, then the burst of LPT instructions.
  // Written once and inlined at its single call site inside the loop.
  int slot_c = 0;                // ring slot of the tile being computed
  int slot_i = 0;                // ring slot the next DMA goes to
  for (int kc = kt0 - (NST - 1); kc < kt1; ++kc) {
    const int kl = kc + NST - 1;             // tile whose DMA is issued in this iteration
    if (kc >= kt0) {
      // tile kc must have landed: tiles kc+1 .. min(kc+NST-2, kt1-1) may still be in flight
      const int ahead = min(kt1 - 1 - kc, NST - 2);
      if (NST >= 4 && ahead >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * LPT>();
      else if (NST >= 3 && ahead >= 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * LPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();          // everyone's part of tile kc landed; compute(kc-1) finished everywhere
    }
#ifdef DSL_ABLATE_BUILD
    if (kl < kt1 && !((p.dbg & 1) && kc >= kt0)) {
#else
    if (kl < kt1) {
#endif
      unsigned char* stage = smem + slot_i * STAGE;
      // every lane decodes ONE pixel row of the stage (row = lane) and the DMA instructions pick their rows'
      // source offsets up with a lane shuffle: one decode per stage instead of one per DMA instruction
      int my_off = -1;             // element offset of this row's source pixel (channel ci0), -1 = zero line
      {
        const int gp = kl * KS + lane;
        if (lane < KS && gp < totpx) {
          const SegSel t = seg_select(p, gp);
          const uint32_t q = gp - t.px0;
          const uint64_t m1 = ((uint64_t)t.m1hi << 32) | t.m1lo, m2 = ((uint64_t)t.m2hi << 32) | t.m2lo;
          const uint32_t img = (uint32_t)(((uint64_t)q * m1) >> 40);
          const uint32_t rem = q - img * t.d1;
          const uint32_t y = (uint32_t)(((uint64_t)rem * m2) >> 40);
          const uint32_t x = rem - y * t.d2;
          const int sy = (int)y * p.stride + tr - p.pad, sx = (int)x * p.stride + ts - p.pad;
          if ((unsigned)sy < (unsigned)t.sh && (unsigned)sx < (unsigned)t.sw)
            my_off = (int)((t.xoff + ((long long)img * t.sh + sy) * t.sw + sx) * p.ldx + ci0);
        }
      }
      int gx[LX];
#pragma unroll
      for (int i = 0; i < LX; ++i) gx[i] = __shfl(my_off, xrow[i], 64);
#pragma unroll
      for (int i = 0; i < LY; ++i) {
        const int gp = kl * KS + yrow[i];
        const gptr_t g = (gp < totpx && co0 + ych[i] < p.cy) ? (gptr_t)(dy_p + (long long)gp * p.cy + co0 + ych[i]) : zero;
        __builtin_amdgcn_global_load_lds(g, (lptr_t)(stage + (wave + NW * i) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < LX; ++i) {
        // keep the select in a named variable: passing the ?: expression straight into the builtin makes
        // hipcc silently drop this kernel's host stub
        const gptr_t g = gx[i] >= 0 ? (gptr_t)(x_p + gx[i] + xch[i]) : zero;
        __builtin_amdgcn_global_load_lds(g, (lptr_t)(stage + TILE_Y + (wave + NW * i) * 1024), 16, 0, 0);
      }
    }
    if (kl >= kt0) slot_i = (slot_i + 1 == NST) ? 0 : slot_i + 1;
    if (kc >= kt0) {
#ifdef DSL_ABLATE_BUILD
      if (!(p.dbg & 2))
#endif
      compute(slot_c);
      if (do_db) {              // this stage's dY tile: rows db_rg * DB_ROWS .. of column pair db_cp (same swizzle as the DMA wrote)
        const unsigned st = lds_base + slot_c * STAGE;
#pragma unroll
        for (int r4 = 0; r4 < DB_ROWS; r4 += 4) {
          unsigned v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = db_rg * DB_ROWS + r4 + j;
            v[j] = lds_read_b32_asm(st + row * YB + ((((db_cp * 4) >> 6) ^ (row & 3)) << 6) + ((db_cp * 4) & 63));
          }
          wait_lds4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            db_lo += __uint_as_float(v[j] << 16);
            db_hi += __uint_as_float(v[j] & 0xffff0000u);
          }
        }
      }
      slot_c = (slot_c + 1 == NST) ? 0 : slot_c + 1;
    }
  }
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 4) return;
#endif

  if (do_db) {               // fold the row groups in a fixed order; one value per column leaves the workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[(db_rg * DB_PAIRS + db_cp) * 2] = db_lo;
    red[(db_rg * DB_PAIRS + db_cp) * 2 + 1] = db_hi;
    __syncthreads();
    if (tid < BCO) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < DB_RG; ++r) sacc += red[(r * DB_PAIRS + (tid >> 1)) * 2 + (tid & 1)];
      const int co = co0 + tid;
      if (p.direct) {
        if (co < p.cd) db_p[co] = sacc;
      } else {
        p.dbws[((long long)sp * p.group + member) * p.cyp + co] = sacc;
      }
    }
  }
  const int frow = lane & 31, fhalf = lane >> 5;
  if (p.direct) {          // one split: this tile is the whole sum - scale and store it into dW, no partial / reduce pass
    float* dw_p = p.dwv[0];
    const float* sc_p = p.scalev[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) {
      dw_p = member == g ? p.dwv[g] : dw_p;
      sc_p = member == g ? p.scalev[g] : sc_p;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
          if (co < p.cd) dw_p[(long long)co * p.krow + col] = sc_p ? acc[ct][it][j] * sc_p[co] : acc[ct][it][j];
        }
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[(((long long)sp * p.group + member) * p.cyp + co) * p.krow + col] = acc[ct][it][j];
      }
    }
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_glds_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wgrad_glds_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x, smem);
}

// ================================================================================================
// v3 weight gradient ("wgrad_pipe"): the same tiles, operands and outputs as wgrad_glds_body, re-scheduled so that nothing
// but the MFMA stream is on the critical path (v2 measured on the head shape, tools/ablate_wgrad.py: DMA-only 50 us +
// MFMA-only 53 us = 87 us together - the two did not overlap at all: every wave issued its whole DMA burst, then decoded
// the next stage's pixels, then waited for its first fragments with an idle matrix pipe):
//   * no per-stage pixel decode: a per-geometry table of 8-byte pixel descriptors {source pixel of tap (0,0), row pitch,
//     per-axis tap validity bits} (PixDesc, built once per geometry on the host, cached by the library) is itself DMA'd
//     into a small LDS ring a few stages ahead; a gather row's offset is then 5 VALU instructions, a dY row's offset is a
//     running counter (dY rows are contiguous in the pixel index); padding, ragged tails and dead stages are the buffer
//     out-of-range rule (zeros land in LDS), so every stage issues the same number of DMA instructions;
//   * KS-pixel stages in an NST-deep ring (32-pixel stages: 4 x 32 KB for the 256x256 tile), stage s+NST-1 is fetched while
//     stage s feeds the MFMAs: its DMA instructions are issued ONE AT A TIME between the MFMAs of the stage (a burst blocks
//     the wave on issue for ~1000 cycles with an empty matrix pipe);
//   * fragment reads of k-step kk+1 are in flight during the MFMAs of step kk, the next stage's first fragments are issued
//     right behind the stage's single barrier, in front of its last MFMA block: the barrier sits inside the MFMA stream.
// vmcnt bookkeeping (P DMA instructions per wave per stage, returned in order): at the barrier that ends stage s, stage
// s+1 must have landed; it was issued during stage s-NST+2, so (NST-3) whole stages plus the Pa pieces of stage s issued so
// far may stay in flight.  Everything issued during stage s-NST+2 or earlier has then landed, including the descriptors
// fetched then: descriptors of stage t are fetched during stage t-(2*NST-2) and read (into registers) right behind the
// barrier that ends stage t-NST, for the gather DMAs issued during stage t-NST+1.
// ================================================================================================
struct PixDesc {
  int32_t base;        // source pixel index of tap (0,0) of this output pixel (may be "virtual": outside the image)
  uint32_t info;       // (source row pitch in pixels) << 16 | x-tap validity bits << 8 | y-tap validity bits
};

__device__ __forceinline__ u32x2 lds_read_b64_asm(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void pin2(u32x2& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin1(unsigned& a) { asm volatile("" : "+v"(a)); }

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__device__ __forceinline__ void wgrad_pipe_body(const WgK& p, const int bid, unsigned char* smem) {
  constexpr int NW = WCO * WCI;
  constexpr int YB = BCO * 2, XB = BCI * 2;
  constexpr int TILE_Y = KS * YB, TILE_X = KS * XB, STAGE = TILE_Y + TILE_X;
  constexpr int NY = TILE_Y / 1024, NX = TILE_X / 1024;
  constexpr int LY = NY / NW, LX = NX / NW;
  constexpr int NDSC = KS * 8 / 256;                            // 256-byte descriptor DMAs per stage
  constexpr int P = NDSC + LY + LX;                              // DMA instructions per wave per stage
  constexpr int CT = BCO / WCO / 32, IT = BCI / WCI / 32, NM = CT * IT;
  constexpr int KK = KS / 16;
  constexpr int DR = 16;                                         // descriptor ring depth (stages)
  constexpr int DESC_BASE = NST * STAGE;
  constexpr int DLEAD = 2 * NST - 2;                             // descriptors run this many stages ahead of the stage computed
  static_assert(NY % NW == 0 && NX % NW == 0 && LY >= 1 && LX >= 1, "tile / wave mismatch");
  static_assert(YB >= 256 && XB >= 256, "64-byte-chunk swizzle needs >= 4 chunks per row");
  static_assert(KK == 2 || KK == 4, "stage depth");
  static_assert(NST >= 3 && DLEAD < DR && NDSC >= 1, "ring depths");
  static_assert((NST - 1) * P + DLEAD * NDSC <= 63, "vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WCI, wave_ci = wave % WCI;
  const int tiles_per_member = p.gx * p.gy;
  const int tiles_per_split_wg = tiles_per_member * p.group;
  const int xcd = bid & 7, jj = bid >> 3;
  const int witem = xcd * p.chunk + jj;
  if (jj >= p.chunk || witem >= tiles_per_split_wg * p.splits) return;
  const int sp = witem / tiles_per_split_wg;
  const int rem_sp = witem - sp * tiles_per_split_wg;
  const int member = rem_sp / tiles_per_member;
  const int rem_wg = rem_sp - member * tiles_per_member;
  const uint16_t* dy_p = p.dyv[0];
  const uint16_t* x_p = p.xv[0];
  float* db_p = p.dbv[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dy_p = member == g ? p.dyv[g] : dy_p;
    x_p = member == g ? p.xv[g] : x_p;
    db_p = member == g ? p.dbv[g] : db_p;
  }
  const int co0 = (rem_wg % p.gx) * BCO;
  const int colt = rem_wg / p.gx;
  const bool do_db = colt == 0 && ((p.dbmask >> member) & 1);
  constexpr int DB_PAIRS = BCO / 2, DB_RG = 64 * NW / DB_PAIRS, DB_ROWS = KS / DB_RG;
  static_assert(DB_ROWS >= 1 && DB_ROWS <= 8 && DB_RG * DB_PAIRS == 64 * NW, "bias-gradient thread mapping");
  const int db_cp = tid % DB_PAIRS, db_rg = tid / DB_PAIRS;
  float db_lo = 0.f, db_hi = 0.f;
  const int ctiles = p.cs / BCI;
  const int tap = colt / ctiles;
  const int ci0 = (colt - tap * ctiles) * BCI;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  if (kt0 >= kt1) return;      // (cannot happen with the host's split factors; a workgroup without stages has nothing to add)

  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)dy_p, 0, (int)p.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_p, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.pixtab, 0, (int)p.pixtab_bytes, 0x00020000);

  // per DMA instruction this lane's (row, source channel) inside the tile (as in wgrad_glds_body)
  unsigned yv[LY];             // running byte offset of this lane's dY element (row of the stage being fetched)
  unsigned xcol[LX], xdaddr[LX];
#pragma unroll
  for (int i = 0; i < LY; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / YB, inrow = off % YB;
    const int ch = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
    yv[i] = co0 + ch < p.cy ? (unsigned)(((kt0 * KS + row) * p.cy + co0 + ch) * 2) : 0x80000000u;
  }
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / XB, inrow = off % XB;
    const int ch = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
    xcol[i] = (unsigned)(ci0 + ch) * 2u;
    xdaddr[i] = (unsigned)(DESC_BASE + row * 8);
  }
  const unsigned y_step = (unsigned)(KS * p.cy * 2);
  const unsigned ldx2 = (unsigned)(p.ldx * 2);
  const unsigned sel = (1u << tr) | (0x100u << ts);

  f32x16 acc[CT][IT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < IT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);
  const int ccol_l = iblk + 4 * (l16 & 3);
  const unsigned lds_base = (unsigned)(size_t)smem;
  unsigned a_off[CT], b_off[IT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = wave_co * (32 * CT) + ct * 32 + ccol_l;
    a_off[ct] = krow_l * YB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int col = wave_ci * (32 * IT) + it * 32 + ccol_l;
    b_off[it] = TILE_Y + krow_l * XB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
  Frag fa[2][CT], fb[2][IT];
  constexpr int NR = 2 * (CT + IT);         // fragment reads per k-step
  // fragment column addresses of the stage being READ (one v_add per column and stage; the k-step / half offsets are immediates)
  unsigned ra[CT], rb[IT];
  auto set_read_stage = [&](unsigned stage_addr) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) ra[ct] = stage_addr + a_off[ct];
#pragma unroll
    for (int it = 0; it < IT; ++it) rb[it] = stage_addr + b_off[it];
  };
  auto rd = [&](int kk, int f, int j) {      // read j of the k-step's NR (kk, j are constants after unrolling)
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 8) return;
#endif
    if (j < 2 * CT) {
      const int ct = j >> 1;
      if (j & 1) fa[f][ct].hi = lds_tr_read_kh<YB, KK>(ra[ct], kk, 1); else fa[f][ct].lo = lds_tr_read_kh<YB, KK>(ra[ct], kk, 0);
    } else {
      const int it = (j - 2 * CT) >> 1;
      if (j & 1) fb[f][it].hi = lds_tr_read_kh<XB, KK>(rb[it], kk, 1); else fb[f][it].lo = lds_tr_read_kh<XB, KK>(rb[it], kk, 0);
    }
  };
  auto issue = [&](int kk, int f) {
#pragma unroll
    for (int j = 0; j < NR; ++j) rd(kk, f, j);
  };
  auto wait_lds = [&](int f) {       // every outstanding LDS read of this wave has landed; the registers it wrote change HERE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { pin2(fa[f][ct].lo); pin2(fa[f][ct].hi); }
#pragma unroll
    for (int it = 0; it < IT; ++it) { pin2(fb[f][it].lo); pin2(fb[f][it].hi); }
  };

  // ---- DMA pieces of one stage: [0, NDSC) descriptors of stage t_desc, [NDSC, NDSC+LY) dY rows, then the gather rows of
  // stage t_data into ring slot ld_slot
  int t_data = kt0, t_desc = kt0, ld_slot = 0;
  unsigned xv[LX];                   // gather offsets of stage t_data (from its descriptors)
  u32x2 dreg[LX];
  auto piece = [&](int k) {          // k is a constant after unrolling
#ifdef DSL_ABLATE_BUILD
    if ((p.dbg & 1) && t_data >= kt0 + NST) return;      // no DMA after the ring's first fill
#endif
    if (k < NDSC) {
      const unsigned v = (unsigned)(t_desc * (KS * 8) + k * 256 + lane * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lptr_t)(smem + DESC_BASE + (t_desc & (DR - 1)) * (KS * 8) + k * 256), 4, v, 0, 0, 0);
    } else if (k < NDSC + LY) {
      const int i = k - NDSC;
      const unsigned v = t_data < kt1 ? yv[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lptr_t)(smem + ld_slot * STAGE + (wave + NW * i) * 1024), 16, v, 0, 0, 0);
    } else {
      const int i = k - NDSC - LY;
      const unsigned v = t_data < kt1 ? xv[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(smem + ld_slot * STAGE + TILE_Y + (wave + NW * i) * 1024), 16, v, 0, 0, 0);
    }
  };
  auto desc_read = [&]() {           // descriptors of stage t_data (landed and barrier-published) -> registers
#pragma unroll
    for (int i = 0; i < LX; ++i) dreg[i] = lds_read_b64_asm(lds_base + xdaddr[i] + (unsigned)((t_data & (DR - 1)) * (KS * 8)));
  };
  auto desc_use = [&]() {            // ... -> this lane's gather offsets (after the wait that covers desc_read)
#pragma unroll
    for (int i = 0; i < LX; ++i) {
      pin2(dreg[i]);
      const unsigned info = dreg[i][1];
      const unsigned px = (unsigned)((int)dreg[i][0] + tr * (int)(info >> 16) + ts);
      xv[i] = (info & sel) == sel ? px * ldx2 + xcol[i] : 0x80000000u;
    }
  };
  auto advance = [&]() {             // the stage's pieces are all issued
    ++t_data;
    ++t_desc;
    ld_slot = (ld_slot + 1 == NST) ? 0 : ld_slot + 1;
#pragma unroll
    for (int i = 0; i < LY; ++i) yv[i] += y_step;
  };
  // one k-step's MFMAs with the DMA pieces [lo, hi) of the stage issued between them: one piece behind every second MFMA
  // (everything here is pinned in source order)
  auto block = [&](int f, int lo, int hi, int rd_kk) {
    bf16x8 a[CT], b[IT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fa[f][ct].lo;
      u.s.h = fa[f][ct].hi;
      a[ct] = u.v;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fb[f][it].lo;
      u.s.h = fb[f][it].hi;
      b[it] = u.v;
    }
    __builtin_amdgcn_sched_barrier(0);
    int k = lo;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#ifdef DSL_ABLATE_BUILD
      if (!(p.dbg & 2))
#endif
      acc[m / IT][m % IT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m / IT], b[m % IT], acc[m / IT][m % IT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the next k-step's fragment reads ride between the MFMAs (a burst of NR reads in front of the block keeps the wave
      // on LDS issue for as long as the block's MFMAs take: measured, the two simply added up)
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (j >= m * NR / NM && j < (m + 1) * NR / NM) rd(rd_kk, f ^ 1, j);
      if ((m & 1) == 1 && k < hi) {
        piece(k);
        ++k;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < P; ++j)
      if (k + j < hi) piece(k + j);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int PA = (KK - 1) * P / KK;          // pieces issued before the stage's barrier (k-steps 0 .. KK-2)

  // ---- prologue: the descriptors of the first NST-1 stages, then NST-1 whole stages (each with the descriptor pieces of a
  // later stage, so that every stage - prologue or not - is exactly P DMA instructions: the waits below count in stages)
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) {
#pragma unroll
    for (int k = 0; k < NDSC; ++k) piece(k);
    ++t_desc;
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) {
    desc_read();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    desc_use();
#pragma unroll
    for (int k = 0; k < P; ++k) piece(k);
    advance();
  }
  // steady state from here: stage s fetches the descriptors of stage s + DLEAD (t_desc) and the data of stage s + NST - 1 (t_data)
  wait_vmcnt<(NST - 2) * P>();                   // stage kt0 landed, and the descriptors fetched with it (stage kt0 + NST - 1's)
  __builtin_amdgcn_s_barrier();
  desc_read();
  set_read_stage(lds_base);
  issue(0, 0);

  unsigned dbr[8];
  int slot_c = 0;
  for (int s = kt0; s < kt1; ++s) {
    const unsigned st = lds_base + slot_c * STAGE;
    const int nslot = (slot_c + 1 == NST) ? 0 : slot_c + 1;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int f = kk & 1;
      wait_lds(f);
      if (kk == 0) desc_use();
      if (kk == 1 && do_db) {          // the dY column sums of this stage (issued in k-step 0)
#pragma unroll
        for (int j = 0; j < DB_ROWS; ++j) {
          pin1(dbr[j]);
          db_lo += __uint_as_float(dbr[j] << 16);
          db_hi += __uint_as_float(dbr[j] & 0xffff0000u);
        }
      }
      if (kk == 0 && do_db) {
#pragma unroll
        for (int j = 0; j < DB_ROWS; ++j) {
          const int row = db_rg * DB_ROWS + j;
          dbr[j] = lds_read_b32_asm(st + row * YB + ((((db_cp * 4) >> 6) ^ (row & 3)) << 6) + ((db_cp * 4) & 63));
        }
      }
      if (kk == KK - 1) {
        wait_vmcnt<(NST - 3) * P + PA>();      // stage s+1 landed (and every older DMA of this wave)
#ifdef DSL_ABLATE_BUILD
        if (!(p.dbg & 16))
#endif
        __builtin_amdgcn_s_barrier();          // ... for every wave; every wave is done reading stage s
      }
      __builtin_amdgcn_sched_barrier(0);
      // this k-step's MFMAs, between them the fragment reads of the next k-step (the last k-step: of the next stage's first,
      // behind the barrier above) and this k-step's share of the stage's DMA pieces
      if (kk < KK - 1) {
        block(f, kk * P / KK, (kk + 1) * P / KK, kk + 1);
      } else {
        set_read_stage(lds_base + nslot * STAGE);
        block(f, kk * P / KK, P, 0);
      }
      if (kk == KK - 1) {
        advance();
        desc_read();                           // descriptors of the stage fetched next (published by the barrier above)
      }
    }
    slot_c = nslot;
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 4) return;
#endif

  if (do_db) {               // fold the row groups in a fixed order; one value per column leaves the workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[(db_rg * DB_PAIRS + db_cp) * 2] = db_lo;
    red[(db_rg * DB_PAIRS + db_cp) * 2 + 1] = db_hi;
    __syncthreads();
    if (tid < BCO) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < DB_RG; ++r) sacc += red[(r * DB_PAIRS + (tid >> 1)) * 2 + (tid & 1)];
      const int co = co0 + tid;
      if (p.direct) {
        if (co < p.cd) db_p[co] = sacc;
      } else {
        p.dbws[((long long)sp * p.group + member) * p.cyp + co] = sacc;
      }
    }
  }
  const int frow = lane & 31, fhalf = lane >> 5;
  if (p.direct) {
    float* dw_p = p.dwv[0];
    const float* sc_p = p.scalev[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) {
      dw_p = member == g ? p.dwv[g] : dw_p;
      sc_p = member == g ? p.scalev[g] : sc_p;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
          if (co < p.cd) dw_p[(long long)co * p.krow + col] = sc_p ? acc[ct][it][j] * sc_p[co] : acc[ct][it][j];
        }
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[(((long long)sp * p.group + member) * p.cyp + co) * p.krow + col] = acc[ct][it][j];
      }
    }
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // persistent form (grid < work items, a multiple of 8 so that a block keeps its XCD): the launch never holds more CUs than
  // its workgroup budget, whatever the number of tiles and splits - the caller's chain of small launches keeps the rest
  const int total = p.chunk * 8;
  for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb, smem);
    __syncthreads();
  }
}

// Several weight-gradient launches of ONE tile configuration as one grid (dsl_conv2d_wgrad_multi): sub-launch s owns the
// blocks [wg_end[s-1], wg_end[s]) (multiples of 8, so a block's XCD is the same as in a launch of its own); its WgK comes
// from a table in device memory, read once with scalar loads before the K loop.  The host orders the sub-launches by
// decreasing work per workgroup: the hardware dispatches blocks in index order, so the short ones fill the tail.
constexpr int kMaxMulti = DSL_MAX_MULTI;
struct WgMultiHdr {
  int nsub;
  int wg_end[kMaxMulti];
};
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_glds_multi_kernel(const WgMultiHdr h, const WgK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int sub = 0, start = 0;
#pragma unroll
  for (int s = 1; s < kMaxMulti; ++s) {
    const bool in = s < h.nsub && (int)blockIdx.x >= h.wg_end[s - 1];
    sub = in ? s : sub;
    start = in ? h.wg_end[s - 1] : start;
  }
  const WgK p = tab[sub];
  wgrad_glds_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x - start, smem);
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_kernel(const WgMultiHdr h, const WgK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int sub = 0, start = 0;
#pragma unroll
  for (int s = 1; s < kMaxMulti; ++s) {
    const bool in = s < h.nsub && (int)blockIdx.x >= h.wg_end[s - 1];
    sub = in ? s : sub;
    start = in ? h.wg_end[s - 1] : start;
  }
  const WgK p = tab[sub];
  wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x - start, smem);
}

// persistent form of the multi launch: `grid` (a multiple of 8) workgroups walk the block list
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_persist_kernel(const WgMultiHdr h, const WgK* __restrict__ tab, int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
    int sub = 0, start = 0;
#pragma unroll
    for (int s = 1; s < kMaxMulti; ++s) {
      const bool in = s < h.nsub && vb >= h.wg_end[s - 1];
      sub = in ? s : sub;
      start = in ? h.wg_end[s - 1] : start;
    }
    const WgK p = tab[sub];
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb - start, smem);
    __syncthreads();
  }
}

// scheduled form (round 4, wgrad_plan_*): the host assigns every valid virtual block to a workgroup (longest-processing-time
// first inside the block's XCD class), sched[r * gridDim.x + b] = the r-th block of workgroup b or -1; what a workgroup computes
// for a block, and hence every result, is the same as in the stride form
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_sched_kernel(const WgMultiHdr h, const WgK* __restrict__ tab,
                                                                               const short* __restrict__ sched, int rounds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int r = 0; r < rounds; ++r) {
    const int vb = __builtin_amdgcn_readfirstlane((int)sched[r * (int)gridDim.x + (int)blockIdx.x]);
    if (vb < 0) break;
    int sub = 0, start = 0;
#pragma unroll
    for (int s = 1; s < kMaxMulti; ++s) {
      const bool in = s < h.nsub && vb >= h.wg_end[s - 1];
      sub = in ? s : sub;
      start = in ? h.wg_end[s - 1] : start;
    }
    const WgK p = tab[sub];
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb - start, smem);
    __syncthreads();
  }
}

// the reduce passes of a multi launch: entry e (one member of one sub-launch with more than one split) owns the blocks
// [blk_start, blk_start + nblk)
struct RedEnt {
  const float* ws;         // this member's first partial: ws + member * cyp * krow
  float* dw;
  const float* scale;
  float* db;               // bias gradient: the splits' column-sum partials dbws[sp * dbstride + c] folded in order (or NULL)
  const float* dbws;
  long long dbstride;
  long long krow, sstride;
  int splits, cd, blk_start, nblk;
};
__global__ void wgrad_reduce_multi_kernel(const RedEnt* __restrict__ tab, int n) {
  int e = 0;
  for (int i = 1; i < n; ++i) e = (int)blockIdx.x >= tab[i].blk_start ? i : e;
  const RedEnt r = tab[e];
  const int lb = (int)blockIdx.x - r.blk_start;
  if (lb == 0 && r.db)
    for (int c = threadIdx.x; c < r.cd; c += blockDim.x) {
      float sacc = 0.f;
      for (int sp = 0; sp < r.splits; ++sp) sacc += r.dbws[sp * r.dbstride + c];
      r.db[c] = sacc;
    }
  const long long total4 = (long long)r.cd * r.krow / 4;
  for (long long i = (long long)lb * blockDim.x + threadIdx.x; i < total4; i += (long long)r.nblk * blockDim.x) {
    const long long el = i * 4;
    const int co = (int)(el / r.krow);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = r.ws + el;
    int sp = 0;
    for (; sp + 4 <= r.splits; sp += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + sp * r.sstride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(base + (sp + 1) * r.sstride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(base + (sp + 2) * r.sstride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(base + (sp + 3) * r.sstride);
      s += (a + b) + (c + d);
    }
    for (; sp < r.splits; ++sp) s += *reinterpret_cast<const f32x4*>(base + sp * r.sstride);
    if (r.scale) s *= r.scale[co];
    *reinterpret_cast<f32x4*>(r.dw + el) = s;
  }
}

struct RedK {
  float* dw[DSL_MAX_GROUP];
  const float* scale[DSL_MAX_GROUP];
  float* db[DSL_MAX_GROUP];        // bias-gradient vectors: summed from dbws, or cleared for the column-sum kernel that follows (or NULL)
  const float* dbws;               // [split][member][cy] column-sum partials of the DMA kernels (NULL: v1 kernel)
};

// sums the split partials ws[split][member][cy][krow] of member blockIdx.y into its dW (x scale)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, const RedK r, int splits, int group, int cy, int cd,
                                    long long krow) {
  const int member = blockIdx.y;
  float* __restrict__ dw = r.dw[0];
  const float* __restrict__ scale = r.scale[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dw = member == g ? r.dw[g] : dw;
    scale = member == g ? r.scale[g] : scale;
  }
  if (blockIdx.x == 0) {
    float* db = r.db[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) db = member == g ? r.db[g] : db;
    if (db)
      for (int c = threadIdx.x; c < cd; c += blockDim.x) {
        float sacc = 0.f;
        if (r.dbws)                 // in-kernel column sums: fold the splits in order; else cleared for the column-sum pass
          for (int sp = 0; sp < splits; ++sp) sacc += r.dbws[((long long)sp * group + member) * cy + c];
        db[c] = sacc;
      }
  }
  const long long total4 = (long long)cd * krow / 4;
  const long long sstride = (long long)group * cy * krow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int co = (int)(e / krow);
    const long long k = e - (long long)co * krow;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = ws + ((long long)member * cy + co) * krow + k;
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {      // 4 independent loads in flight per thread
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + sp * sstride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(base + (sp + 1) * sstride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(base + (sp + 2) * sstride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(base + (sp + 3) * sstride);
      s += (a + b) + (c + d);
    }
    for (; sp < splits; ++sp) s += *reinterpret_cast<const f32x4*>(base + sp * sstride);
    if (scale) s *= scale[co];
    *reinterpret_cast<f32x4*>(dw + e) = s;
  }
}

}  // namespace

// ---- v3 weight gradient: per-geometry pixel descriptor tables (PixDesc).  The table lives in CALLER-OWNED device memory
// (dsl_wgrad_desc.pixtab, >= dsl_wgrad_pixtab_bytes): dsl_wgrad_pixtab_fill writes it with a kernel on the caller's stream, once per
// geometry; the library allocates nothing and copies nothing (until round 6 it kept a process-lifetime hipMalloc'd cache here)
namespace {
constexpr int kWgV3KS = 32, kWgV3DR = 16;
struct PixGeo {
  int nseg, n, stride, pad, kh, kw;
  int gh[DSL_MAX_SEG], gw[DSL_MAX_SEG], sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];
  int pxstart[DSL_MAX_SEG + 1];
  long long xoff[DSL_MAX_SEG];
};
__global__ void pixtab_fill_kernel(PixDesc* __restrict__ out, PixGeo g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.pxstart[g.nseg]) return;
  int s = 0;
#pragma unroll
  for (int t = 1; t < DSL_MAX_SEG; ++t)
    if (t < g.nseg && i >= g.pxstart[t]) s = t;
  const int gh = g.gh[s], gw = g.gw[s], sh = g.sh[s], sw = g.sw[s];
  int r = i - g.pxstart[s];
  const int img = r / (gh * gw);
  r -= img * gh * gw;
  const int y = r / gw, x = r - y * gw;
  const int y0 = y * g.stride - g.pad, x0 = x * g.stride - g.pad;
  unsigned ym = 0, xm = 0;
  for (int a = 0; a < g.kh; ++a) if ((unsigned)(y0 + a) < (unsigned)sh) ym |= 1u << a;
  for (int c = 0; c < g.kw; ++c) if ((unsigned)(x0 + c) < (unsigned)sw) xm |= 1u << c;
  PixDesc e;
  e.base = (int32_t)(g.xoff[s] + ((long long)img * sh + y0) * sw + x0);
  e.info = ((unsigned)sw << 16) | (xm << 8) | ym;
  out[i] = e;
}

int wgrad_slots();
bool wgrad_persist() {
  // measured (tools/exp_env.sh, bench.py N = 2): persistent grids of 128 workgroups +2.3 % (96 .. 160 within 0.3 %, 64: -1 %)
  return true;
}
bool wgrad_v3_enabled() {
  return true;
}
// ring depth of the v3 kernel per tile configuration (1: 256x256 -> 4 x 32 KB; 2, 3: 24 KB stages)
int wgrad_v3_nst(int cfg) { return cfg == 1 ? 4 : 5; }
bool wgrad_v3_ok(const dsl_wgrad_desc* d, int cfg) {
  if (!wgrad_v3_enabled() || cfg < 1 || cfg > 3 || d->kh > 8 || d->kw > 8) return false;
  long long px = 0, xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    px += (long long)d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
    if (d->sw[s] >= 65536) return false;
  }
  const long long ldx = d->ldx > 0 ? d->ldx : d->cs;
  return px * d->cy * 2 < 0x7fff0000LL && xo * ldx * 2 < 0x7fff0000LL && px < (1 << 20);
}
int wgrad_v3_fill(const dsl_wgrad_desc* d, WgK& k, long long px) {
  const size_t need = (size_t)px * sizeof(PixDesc);
  DSL_CHECK(d->pixtab != nullptr && d->pixtab_bytes >= need,
            "dsl_conv2d_wgrad: this geometry runs the pipelined kernel and needs dsl_wgrad_desc.pixtab (%zu bytes given, %zu needed: "
            "dsl_wgrad_pixtab_bytes / dsl_wgrad_pixtab_fill)", d->pixtab ? d->pixtab_bytes : (size_t)0, need);
  k.pixtab = d->pixtab;
  k.pixtab_bytes = (unsigned)need;
  k.ybytes = (unsigned)(px * d->cy * 2);
  return 0;
}
size_t wgrad_v3_lds(int cfg) {
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128};
  return (size_t)wgrad_v3_nst(cfg) * kWgV3KS * 2 * (bcos[cfg] + bcis[cfg]) + (size_t)kWgV3DR * kWgV3KS * 8;
}
}  // namespace

// wgrad tile configurations: 0 = v1 (BCO 128|64 x 128, register staged), 1 = 256x256, 2 = 256co x 128ci,
// 3 = 128co x 256ci, 4 = 128x128 (v2)
static int wgrad_pick(const dsl_wgrad_desc* d) {
  const int force = d->splits < 0 ? -d->splits : 0;       // test hook: splits = -(cfg+1) forces a config
  if (force) return force - 1;
  if (d->cy % 128) return d->cs % 256 == 0 ? 3 : 4;      // cy = 64 (mod 128): the 128-cout tiles, upper half reads zeros
  if (d->cy % 256 == 0 && d->cs % 256 == 0) return 1;
  if (d->cy % 256 == 0) return 2;
  if (d->cs % 256 == 0) return 3;
  return 4;
}
static int wgrad_geometry(const dsl_wgrad_desc* d, int* ktiles, int* tiles, int* bco) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  *ktiles = (int)((px + 63) / 64);
  const int cfg = wgrad_pick(d);
  const int bcos[5] = {(d->cy % 128 == 0) ? 128 : 64, 256, 256, 128, 128};
  const int bcis[5] = {128, 256, 128, 256, 128};
  *bco = bcos[cfg];
  *tiles = ((d->cy + *bco - 1) / *bco) * (d->kh * d->kw * d->cs / bcis[cfg]);
  return cfg;
}

static int wgrad_splits_for(const dsl_wgrad_desc* d, int count) {
  int ktiles, tiles, bco;
  const int cfg = wgrad_geometry(d, &ktiles, &tiles, &bco);
  tiles *= count;
  const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;    // at least 4 K stages per split
  int splits;
  if (cfg == 0) {
    splits = (768 + tiles - 1) / tiles;                    // v1: 2-3 small workgroups per CU
  } else {
    const int per_cu = cfg == 4 ? 2 : 1;                   // 128x128 tiles: two workgroups per CU
    // one full round, never a nearly-empty second one.  (Accumulating the split partials with XCD-local L2 float
    // atomics instead of writing them out was measured: 117 vs 85 us on the head shape - L2 atomics retire about
    // two lanes per clock per channel.)
    // option wgrad_slots < 256 leaves CUs free: the weight gradients run on the side stream under the caller's chain of
    // small convolutions, and a full round of 128 KB-LDS workgroups that live for 100-250 us would leave those
    // kernels only the handful of CUs the round did not cover
    // (measured, bench.py N = 2: 256 -> 305, 224 -> 306, 192 -> 309, 160 -> 313, 128 -> 310 img/s)
    const int slots = d->slots > 0 ? d->slots : wgrad_slots();
    splits = slots * per_cu / tiles;
  }
  if (splits > max_by_k) splits = max_by_k;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  {
    // No EMPTY split (round 4): split i covers stages [i * tps, (i + 1) * tps), tps = ceil(stages / splits); with 9 or more splits
    // and few stages the last ones start past the end - (splits - 1) * tps >= stages, e.g. 129 stages in 16 splits of 9 - their
    // workgroups return without writing their partial tile and the reduce pass adds whatever the scratch buffer held.  The comment
    // "cannot happen with the host's split factors" in the kernels was wrong for this corner; the planner of the multi launches
    // normalises the same way (plan_norm_splits).
    long long px = 0;
    for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
    const int ks = wgrad_v3_ok(d, cfg) ? kWgV3KS : 64;
    const int stages = (int)((px + ks - 1) / ks);
    const int tps = (stages + splits - 1) / splits;
    splits = (stages + tps - 1) / tps;
  }
  return splits;
}

extern "C" int dsl_wgrad_splits(const dsl_wgrad_desc* d) { return wgrad_splits_for(d, 1); }

// Pixel descriptor table of d's geometry (the pipelined kernels' gather addresses): size and fill.  0 bytes: this geometry runs a
// kernel that needs none.
extern "C" size_t dsl_wgrad_pixtab_bytes(const dsl_wgrad_desc* d) {
  if (d == nullptr || d->nseg < 1 || d->nseg > DSL_MAX_SEG) return 0;
  if (!wgrad_v3_ok(d, wgrad_pick(d))) return 0;
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  return (size_t)px * sizeof(PixDesc);
}
extern "C" int dsl_wgrad_pixtab_fill(const dsl_wgrad_desc* d, void* table, size_t bytes, void* stream) {
  DSL_CHECK(d != nullptr && d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_wgrad_pixtab_fill: bad descriptor");
  const size_t need = dsl_wgrad_pixtab_bytes(d);
  if (need == 0) return 0;
  DSL_CHECK(table != nullptr && bytes >= need, "dsl_wgrad_pixtab_fill: table too small (%zu < %zu)", bytes, need);
  PixGeo g;
  memset(&g, 0, sizeof(g));
  g.nseg = d->nseg; g.n = d->n; g.stride = d->stride; g.pad = d->pad; g.kh = d->kh; g.kw = d->kw;
  int px = 0;
  long long xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    g.gh[s] = d->gh[s]; g.gw[s] = d->gw[s]; g.sh[s] = d->sh[s]; g.sw[s] = d->sw[s];
    g.pxstart[s] = px;
    g.xoff[s] = xo;
    px += d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
  }
  for (int s = d->nseg; s <= DSL_MAX_SEG; ++s) g.pxstart[s] = px;
  hipLaunchKernelGGL(pixtab_fill_kernel, dim3((px + 255) / 256), dim3(256), 0, (hipStream_t)stream, (PixDesc*)table, g);
  DSL_CHECK(hipGetLastError() == hipSuccess, "dsl_wgrad_pixtab_fill: launch failed");
  return 0;
}

static size_t wgrad_cy_pad(const dsl_wgrad_desc* d) {       // rows of one partial tile set in the workspace
  int ktiles, tiles, bco;
  wgrad_geometry(d, &ktiles, &tiles, &bco);
  return (size_t)(d->cy + bco - 1) / bco * bco;
}

extern "C" size_t dsl_wgrad_workspace_bytes(const dsl_wgrad_desc* d) {
  const int splits = d->splits > 0 ? d->splits : dsl_wgrad_splits(d);
  return (size_t)splits * wgrad_cy_pad(d) * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);    // + one row of column sums
}

extern "C" size_t dsl_wgrad_group_workspace_bytes(const dsl_wgrad_desc* descs, int count) {
  if (!descs || count < 1) return 0;
  if (count == 1) return dsl_wgrad_workspace_bytes(descs);
  return (size_t)wgrad_splits_for(descs, count) * count * wgrad_cy_pad(descs) * ((size_t)descs->kh * descs->kw * descs->cs + 1) * sizeof(float);
}

extern "C" int dsl_colsum(const void* x, float* out, long rows, int c, int ld, void* stream);
int dsl_colsum_acc(const void* x, float* out, long rows, int c, int ld, void* stream);   // no memset: out += column sums

static bool wgrad_same_geometry(const dsl_wgrad_desc* a, const dsl_wgrad_desc* b) {
  if (a->ldx != b->ldx || a->shared != b->shared) return false;
  if (a->nseg != b->nseg || a->n != b->n || a->cs != b->cs || a->cy != b->cy || a->cd != b->cd || a->kh != b->kh ||
      a->kw != b->kw || a->stride != b->stride || a->pad != b->pad)
    return false;
  for (int s = 0; s < a->nseg; ++s)
    if (a->gh[s] != b->gh[s] || a->gw[s] != b->gw[s] || a->sh[s] != b->sh[s] || a->sw[s] != b->sw[s]) return false;
  return true;
}

// `count` convolutions of one geometry as one launch (count == 1: the plain weight gradient)
static int wgrad_launch(const dsl_wgrad_desc* descs, int count, void* stream) {
  const dsl_wgrad_desc* d = descs;
  DSL_CHECK(d != nullptr && count >= 1 && count <= DSL_MAX_GROUP, "dsl_conv2d_wgrad: bad group (count=%d)", count);
  DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_conv2d_wgrad: nseg=%d", d->nseg);
  DSL_CHECK(d->cs % 128 == 0, "dsl_conv2d_wgrad: Cin=%d must be a multiple of 128", d->cs);
  DSL_CHECK(d->cy % 64 == 0 && d->cd <= d->cy, "dsl_conv2d_wgrad: bad cy=%d cd=%d", d->cy, d->cd);
  for (int g = 0; g < count; ++g) {
    DSL_CHECK(descs[g].dy && descs[g].x && descs[g].dw, "dsl_conv2d_wgrad: null pointer (member %d)", g);
    DSL_CHECK(wgrad_same_geometry(d, &descs[g]), "dsl_conv2d_wgrad_group: member %d has a different geometry", g);
  }
  DSL_CHECK(d->workspace, "dsl_conv2d_wgrad: null workspace");
  int ktiles, tiles, bco;
  const int cfg = wgrad_geometry(d, &ktiles, &tiles, &bco);
  if (cfg == 0 && count > 1) {          // the register-staged kernel has no group form: run the members one by one
    for (int g = 0; g < count; ++g) {
      dsl_wgrad_desc t = descs[g];
      t.workspace = d->workspace;
      t.workspace_bytes = d->workspace_bytes;
      t.splits = 0;
      const int rc = wgrad_launch(&t, 1, stream);
      if (rc) return rc;
    }
    return 0;
  }
  const int splits = count == 1 ? (d->splits > 0 ? d->splits : dsl_wgrad_splits(d)) : wgrad_splits_for(d, count);
  const int cyp = (int)wgrad_cy_pad(d);
  const size_t need = (size_t)splits * count * cyp * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);
  DSL_CHECK(d->workspace_bytes >= need, "dsl_conv2d_wgrad: workspace too small (%zu < %zu)", d->workspace_bytes, need);
  WgK k;
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg; k.n = d->n;
  int px = 0;
  long long xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.gh[s] = d->gh[s]; k.gw[s] = d->gw[s]; k.sh[s] = d->sh[s]; k.sw[s] = d->sw[s];
    k.pxstart[s] = px;
    k.xoff[s] = xo;
    k.dhw[s] = make_fastdiv((uint32_t)(d->gh[s] * d->gw[s]));
    k.dwd[s] = make_fastdiv((uint32_t)d->gw[s]);
    px += d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
  }
  DSL_CHECK(px < (1 << 20), "dsl_conv2d_wgrad: %d pixels exceed the 2^20 fast-division range", px);
  const int ldx = d->ldx > 0 ? d->ldx : d->cs;
  DSL_CHECK(ldx >= d->cs && ldx % 8 == 0, "dsl_conv2d_wgrad: ldx=%d must be >= cs=%d and a multiple of 8", ldx, d->cs);
  DSL_CHECK(xo * ldx < (1LL << 31), "dsl_conv2d_wgrad: X has more than 2^31 elements");
  k.pxstart[d->nseg] = px;
  k.totpx = px;
  k.cs = d->cs; k.cy = d->cy; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.ktiles = ktiles;
  k.tiles_per_split = (ktiles + splits - 1) / splits;
  k.ctiles_per_tap = d->cs / 128;
  k.krow = (long long)d->kh * d->kw * d->cs;
  k.dy = (const uint16_t*)d->dy; k.x = (const uint16_t*)d->x; k.ws = (float*)d->workspace;
  k.group = count;
  k.ldx = ldx;
  k.cyp = cyp;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    k.dyv[g] = (const uint16_t*)descs[g < count ? g : 0].dy;
    k.xv[g] = (const uint16_t*)descs[g < count ? g : 0].x;
  }
  hipStream_t st = (hipStream_t)stream;
  // weight gradient: dY and X read once, dW written once (fp32)
  const int prof = dsl_prof_active()
                       ? dsl_prof_begin(3, 2.0 * count * px * (double)d->cd * d->kh * d->kw * d->cs, st,
                                        count * ((double)px * d->cd * 2.0 + (double)xo * d->cs * 2.0 + (double)d->cd * d->kh * d->kw * d->cs * 4.0))
                       : -1;
  if (cfg >= 1) {
    const int bcis[5] = {128, 256, 128, 256, 128};
    const int bci = bcis[cfg];
    DSL_CHECK(cyp % bco == 0 && d->cs % bci == 0, "dsl_conv2d_wgrad: tile config %d does not divide cy=%d / cs=%d", cfg, d->cy, d->cs);
    k.gx = cyp / bco;
    k.gy = d->kh * d->kw * d->cs / bci;
    k.splits = splits;
    k.dbws = (float*)d->workspace + (size_t)splits * count * cyp * k.krow;      // behind the dW partials
    for (int g = 0; g < count; ++g)
      if (descs[g].db) k.dbmask |= 1 << g;
#ifdef DSL_ABLATE_BUILD
    { const char* e = getenv("DSL_ABLATE"); k.dbg = e ? atoi(e) : 0; }
#endif
    k.chunk = (k.gx * k.gy * count * splits + 7) / 8;
    dim3 grid2(k.chunk * 8);
    const bool v3 = wgrad_v3_ok(d, cfg);
    const int kss[5] = {64, 64, 64, 64, 64}, nsts[5] = {2, 2, 3, 3, 2};
    const int ks = v3 ? kWgV3KS : kss[cfg];
    // the stage length of this tile configuration defines the K-tile unit
    k.ktiles = (px + ks - 1) / ks;
    k.tiles_per_split = (k.ktiles + splits - 1) / splits;
    if (v3)
      if (int rc = wgrad_v3_fill(d, k, px)) return rc;
    const size_t lds2 = v3 ? wgrad_v3_lds(cfg) : (size_t)nsts[cfg] * ks * 2 * (bco + bci);
#define LAUNCHW(KERNEL, A, B, C_, D, KS_, S_)                                                                         \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)KERNEL<A, B, C_, D, KS_, S_>,                                                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((KERNEL<A, B, C_, D, KS_, S_>), grid2, dim3(64 * C_ * D), lds2, st, k);                        \
  } while (0)
    if (v3) {
      if (wgrad_persist()) {
        const int cap = ((d->slots > 0 ? d->slots : wgrad_slots()) + 7) / 8 * 8;
        if ((int)grid2.x > cap) grid2.x = cap;
      }
      switch (cfg) {
        case 1: LAUNCHW(wgrad_pipe_kernel, 256, 256, 2, 4, 32, 4); break;
        case 2: LAUNCHW(wgrad_pipe_kernel, 256, 128, 4, 2, 32, 5); break;
        default: LAUNCHW(wgrad_pipe_kernel, 128, 256, 2, 4, 32, 5); break;
      }
    } else {
      switch (cfg) {
        case 1: LAUNCHW(wgrad_glds_kernel, 256, 256, 2, 4, 64, 2); break;
        case 2: LAUNCHW(wgrad_glds_kernel, 256, 128, 4, 2, 64, 3); break;
        case 3: LAUNCHW(wgrad_glds_kernel, 128, 256, 2, 4, 64, 3); break;
        default: LAUNCHW(wgrad_glds_kernel, 128, 128, 2, 2, 64, 2); break;
      }
    }
#undef LAUNCHW
  } else if (bco == 128) {
    dim3 grid(d->cy / bco, d->kh * d->kw * d->cs / 128, splits);
    const size_t lds = 2 * (size_t)(64 * bco * 2 + 64 * 256);
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<128>), grid, dim3(256), lds, st, k);
  } else {
    dim3 grid(d->cy / bco, d->kh * d->kw * d->cs / 128, splits);
    const size_t lds = 2 * (size_t)(64 * bco * 2 + 64 * 256);
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<64>), grid, dim3(256), lds, st, k);
  }
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("wgrad_kernel");
  const long long total4 = (long long)d->cd * k.krow / 4;
  int rb = (int)((total4 + 255) / 256);
  if (rb > 4096 / count) rb = 4096 / count;
  RedK r;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    r.dw[g] = descs[g < count ? g : 0].dw;
    r.scale[g] = descs[g < count ? g : 0].scale;
    r.db[g] = g < count ? descs[g].db : nullptr;
  }
  r.dbws = cfg >= 1 ? k.dbws : nullptr;
  if (d->shared && count > 1) {
    // the members are applications of ONE convolution (weights shared along a recurrence): their partial tiles are just
    // more splits of the same dW - [split][member] pairs are contiguous in the workspace
    for (int g = 1; g < count; ++g)
      DSL_CHECK(descs[g].dw == d->dw && descs[g].scale == d->scale && !descs[g].db, "dsl_conv2d_wgrad_group: shared members must share dw / scale and have no db");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb * count, 1), dim3(256), 0, st, (const float*)d->workspace, r, splits * count, 1,
                       cyp, d->cd, k.krow);
    DSL_LAUNCH_CHECK("wgrad_reduce_kernel");
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb, count), dim3(256), 0, st, (const float*)d->workspace, r, splits, count,
                     cyp, d->cd, k.krow);
  DSL_LAUNCH_CHECK("wgrad_reduce_kernel");
  if (cfg == 0)              // register-staged kernel: separate column-sum pass (db was cleared by the reduce kernel above)
    for (int g = 0; g < count; ++g)
      if (descs[g].db) {
        const int rc = dsl_colsum_acc(descs[g].dy, descs[g].db, (long)px, d->cd, d->cy, stream);
        if (rc) return rc;
      }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// multi launch: the weight gradients of several geometries (one tile configuration) as ONE grid + ONE reduce grid
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxRed = 128, kMaxColsum = 64, kSchedMax = 8192;
struct ColsumItem { const void* x; float* out; long long rows; int c, ld, clear; };
struct WgMultiTable {
  int magic, cfg, nsub, total_blocks;
  WgMultiHdr hdr;
  int n_red, red_blocks, n_colsum, v3;
  double flops, bytes;
  ColsumItem colsum[kMaxColsum];
  WgK k[kMaxMulti];
  RedEnt red[kMaxRed];
  // scheduled persistent launch (wgrad_plan): sched_grid workgroups, sched_rounds blocks each at most; 0 rounds = stride form
  int sched_grid, sched_rounds;
  int plan_makespan, plan_items;      // (stages incl. the per-item overhead; valid blocks) - what dsl_wgrad_multi_info reports
  short sched[kSchedMax];
};
constexpr int kMultiMagic = 0x574d5431;

int wgrad_fill_k(const dsl_wgrad_desc* descs, int count, int splits, int cfg, bool v3, WgK& k, long long* px_out, long long* xo_out) {
  const dsl_wgrad_desc* d = descs;
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg; k.n = d->n;
  int px = 0;
  long long xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.gh[s] = d->gh[s]; k.gw[s] = d->gw[s]; k.sh[s] = d->sh[s]; k.sw[s] = d->sw[s];
    k.pxstart[s] = px;
    k.xoff[s] = xo;
    k.dhw[s] = make_fastdiv((uint32_t)(d->gh[s] * d->gw[s]));
    k.dwd[s] = make_fastdiv((uint32_t)d->gw[s]);
    px += d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
  }
  DSL_CHECK(px < (1 << 20), "dsl_conv2d_wgrad: %d pixels exceed the 2^20 fast-division range", px);
  const int ldx = d->ldx > 0 ? d->ldx : d->cs;
  DSL_CHECK(ldx >= d->cs && ldx % 8 == 0, "dsl_conv2d_wgrad: ldx=%d must be >= cs=%d and a multiple of 8", ldx, d->cs);
  DSL_CHECK(xo * ldx < (1LL << 31), "dsl_conv2d_wgrad: X has more than 2^31 elements");
  k.pxstart[d->nseg] = px;
  k.totpx = px;
  k.cs = d->cs; k.cy = d->cy; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.ctiles_per_tap = d->cs / 128;
  k.krow = (long long)d->kh * d->kw * d->cs;
  k.group = count;
  k.ldx = ldx;
  k.cyp = (int)wgrad_cy_pad(d);
  k.cd = d->cd;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    const dsl_wgrad_desc& m = descs[g < count ? g : 0];
    k.dyv[g] = (const uint16_t*)m.dy;
    k.xv[g] = (const uint16_t*)m.x;
    k.dwv[g] = m.dw;
    k.scalev[g] = m.scale;
    k.dbv[g] = m.db;
    if (g < count && m.db) k.dbmask |= 1 << g;
  }
  k.dy = k.dyv[0]; k.x = k.xv[0];
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128};
  DSL_CHECK(k.cyp % bcos[cfg] == 0 && d->cs % bcis[cfg] == 0, "dsl_conv2d_wgrad: tile config %d does not divide cy=%d / cs=%d", cfg, d->cy, d->cs);
  k.gx = k.cyp / bcos[cfg];
  k.gy = d->kh * d->kw * d->cs / bcis[cfg];
  k.splits = splits;
  k.chunk = (k.gx * k.gy * count * splits + 7) / 8;
  const int ks = v3 ? kWgV3KS : 64;
  k.ktiles = (px + ks - 1) / ks;
  k.tiles_per_split = (k.ktiles + splits - 1) / splits;
  if (v3)
    if (int rc = wgrad_v3_fill(d, k, px)) return rc;
  *px_out = px;
  *xo_out = xo;
  return 0;
}

int wgrad_slots() {
  const int v = dsl_option("wgrad_slots");
  return v > 0 ? v : 128;
}

// ---- launch planner of the multi launches (round 4) ----------------------------------------------------------------------
// A multi launch is a list of work items (one output tile x one K split) of very different lengths - the FPN's run from 3 to
// 525 64-pixel K tiles - on a persistent grid of <= `cap` workgroups.  Round 3 chose the split factors from one target length
// (total / slots, rounded per sub-launch) and let workgroup b walk the items b, b + G, b + 2G ...: the predictors came out as
// 144 equal items on 128 workgroups (two rounds for 16 of them: 2 x the ideal time), the FPN as 173 items whose second round
// paired the longest with the middle ones (91 units against an ideal 64), layer3 as 2 x 108 half-length items + a reduce pass
// where 108 whole ones fit one round without any partial tile.  The planner simulates what the grid will do: for every
// candidate vector of split factors it assigns the items to workgroups (longest first, each to the least loaded workgroup of
// the item's XCD class, so a block keeps the L2 its neighbours use), takes the longest workgroup's sum (+ a fixed cost per
// item: ring fill and the tile's stores) and adds the reduce pass the split partials would need; the cheapest vector wins and
// its assignment becomes the launch's schedule table.  Results do not depend on the schedule: an item computes the same tile
// from the same stages whoever runs it, and the reduce pass folds the splits in split order.
struct PlanSub { int stages, tiles, max_sp; long long tile_elems; };      // K stages of the kernel's unit, output tiles (all members), elements per tile set
int wgrad_plan_mode() {
  return 1;
}
inline int plan_norm_splits(int stages, int sp) {      // no empty split: sp -> ceil(stages / ceil(stages / sp))
  if (sp < 1) sp = 1;
  const int tps = (stages + sp - 1) / sp;
  return (stages + tps - 1) / tps;
}
// LPT assignment of the valid blocks of the launch (sub-launches in table order) to G workgroups.  Returns the makespan in
// stages (incl. `ovh` per item); sched (may be null) gets G * rounds entries.
long long plan_simulate(const PlanSub* subs, const int* splits, int nsub, int G, int ovh, short* sched, int sched_cap, int* rounds_out,
                        int* items_out) {
  struct It { int cost, vb; };
  std::vector<It> cls[8];
  int base = 0, items = 0;
  for (int i = 0; i < nsub; ++i) {
    const int witems = subs[i].tiles * splits[i];
    const int chunk = (witems + 7) / 8;
    const int tps = (subs[i].stages + splits[i] - 1) / splits[i];
    for (int bid = 0; bid < chunk * 8; ++bid) {
      const int xcd = bid & 7, jj = bid >> 3, w = xcd * chunk + jj;
      if (jj >= chunk || w >= witems) continue;
      const int sp = w / subs[i].tiles;
      const int k0 = sp * tps, k1 = std::min(k0 + tps, subs[i].stages);
      if (k1 <= k0) continue;
      cls[xcd].push_back({ovh + (k1 - k0), base + bid});
      ++items;
    }
    base += chunk * 8;
  }
  const int per = G / 8;
  long long makespan = 0;
  int rounds = 0;
  std::vector<long long> load(G, 0);
  std::vector<std::vector<int>> mine(G);
  for (int x = 0; x < 8; ++x) {
    std::stable_sort(cls[x].begin(), cls[x].end(), [](const It& a, const It& b) { return a.cost > b.cost; });
    for (const It& it : cls[x]) {
      int best = x;
      for (int j = 1; j < per; ++j)
        if (load[x + 8 * j] < load[best]) best = x + 8 * j;
      load[best] += it.cost;
      mine[best].push_back(it.vb);
    }
  }
  for (int b = 0; b < G; ++b) {
    makespan = std::max(makespan, load[b]);
    rounds = std::max(rounds, (int)mine[b].size());
  }
  if (sched) {
    if ((long long)rounds * G > sched_cap) { rounds = 0; }      // does not fit the table: the caller falls back to the stride form
    else {
      for (int i = 0; i < rounds * G; ++i) sched[i] = -1;
      for (int b = 0; b < G; ++b)
        for (size_t r = 0; r < mine[b].size(); ++r) sched[r * G + b] = (short)mine[b][r];
    }
  }
  if (rounds_out) *rounds_out = rounds;
  if (items_out) *items_out = items;
  return makespan;
}
// microseconds per stage / fixed stages per item of a tile configuration (fits of round 3's traces: the towers' direct tiles run
// 1 400 32-pixel stages in 868 us alone; the predictors' 128 x 256 items 198 stages in ~85 us)
inline double plan_stage_us(int cfg) { return cfg == 1 ? 0.62 : 0.43; }
inline int plan_ovh(int cfg) { return cfg == 1 ? 8 : 8; }
struct PlanOut { int splits[kMaxMulti]; int grid, makespan, items; double us; };
void wgrad_plan(const PlanSub* subs, int nsub, int cfg, int cap, PlanOut* out) {
  const int ovh = plan_ovh(cfg);
  long long total = 0;
  int smax = 1;
  for (int i = 0; i < nsub; ++i) { total += (long long)subs[i].stages * subs[i].tiles; smax = std::max(smax, subs[i].stages); }
  // candidate target lengths: every value ceil(stages_i / j) that changes some sub-launch's split factor, within a window around
  // the balanced length, plus "no split at all"
  const long long bal = std::max<long long>(8, (total + cap - 1) / cap);
  std::vector<int> cand;
  cand.push_back(smax);
  for (int i = 0; i < nsub; ++i)
    for (int j = 1; j <= subs[i].max_sp; ++j) {
      const int l = (subs[i].stages + j - 1) / j;
      if (l >= bal / 3 && l <= bal * 4) cand.push_back(l);
    }
  std::sort(cand.begin(), cand.end());
  cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
  double best = 1e30;
  std::vector<std::vector<int>> seen;
  for (int l : cand) {
    std::vector<int> sp(nsub);
    long long items = 0;
    for (int i = 0; i < nsub; ++i) {
      int v = (subs[i].stages + l - 1) / l;
      if (v > subs[i].max_sp) v = subs[i].max_sp;
      sp[i] = plan_norm_splits(subs[i].stages, v);
      items += (long long)sp[i] * subs[i].tiles;
    }
    if (std::find(seen.begin(), seen.end(), sp) != seen.end()) continue;
    seen.push_back(sp);
    // the smallest grid that reaches the best makespan (an XCD class may hold more items than items / 8)
    int G = (int)std::min<long long>(cap, (items + 7) / 8 * 8);
    if (G < 8) G = 8;
    int n_items = 0;
    long long ms = plan_simulate(subs, sp.data(), nsub, cap, ovh, nullptr, 0, nullptr, &n_items);
    {
      int g = G;
      for (; g < cap; g += 8)
        if (plan_simulate(subs, sp.data(), nsub, g, ovh, nullptr, 0, nullptr, nullptr) <= ms) break;
      G = g;
    }
    double red_bytes = 0;
    for (int i = 0; i < nsub; ++i)
      if (sp[i] > 1) red_bytes += (double)(sp[i] + 1) * subs[i].tile_elems * 4.0;      // partials written, read back, dW written
    const double us = ms * plan_stage_us(cfg) + (red_bytes > 0 ? 6.0 + red_bytes / 3.0e6 : 0.0);
    if (us < best) {
      best = us;
      for (int i = 0; i < nsub; ++i) out->splits[i] = sp[i];
      out->grid = G; out->makespan = (int)ms; out->items = n_items; out->us = us;
    }
  }
}
long long wgrad_px(const dsl_wgrad_desc* d) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  return px;
}
bool wgrad_multi_v3(const dsl_wgrad_desc* descs, const int* counts, int nsub, int cfg) {
  bool v3 = true;
  int off = 0;
  for (int s = 0; s < nsub; ++s) { v3 = v3 && wgrad_v3_ok(&descs[off], cfg); off += counts[s]; }
  return v3;
}
// workgroup budget of a multi launch: the library's (DSL_WGRAD_SLOTS, default 128) unless a descriptor asks for its own
// (dsl_wgrad_desc.slots > 0: the launches at the very end of a backward pass, with nothing left to run beside them, take more)
int wgrad_multi_cap(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  int n = 0, cap = 0;
  for (int s = 0; s < nsub; ++s) n += counts[s];
  for (int i = 0; i < n; ++i) cap = std::max(cap, descs[i].slots);
  if (cap <= 0) cap = wgrad_slots();
  return (std::min(cap, 256) + 7) / 8 * 8;
}

// the planner's view of a launch's sub-launches (in the caller's order)
void wgrad_plan_subs(const dsl_wgrad_desc* descs, const int* counts, int nsub, int ks, PlanSub* subs) {
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    const long long px = wgrad_px(&descs[off]);
    subs[s].stages = (int)((px + ks - 1) / ks);
    subs[s].tiles = tiles * counts[s];
    subs[s].max_sp = std::max(1, subs[s].stages / (256 / ks));          // at least 256 pixels of K per split (round 3's rule)
    subs[s].tile_elems = (long long)counts[s] * (long long)wgrad_cy_pad(&descs[off]) * ((long long)descs[off].kh * descs[off].kw * descs[off].cs);
    off += counts[s];
  }
}

// split factors of a multi launch: every workgroup gets at most ~1/slots of the launch's K-tile iterations
int wgrad_multi_splits(const dsl_wgrad_desc* descs, const int* counts, int nsub, int* splits) {
  {
    const int cfg0 = wgrad_pick(descs);
    if (wgrad_plan_mode() && cfg0 >= 1 && cfg0 <= 3 && wgrad_persist() && wgrad_multi_v3(descs, counts, nsub, cfg0)) {
      PlanSub subs[kMaxMulti];
      wgrad_plan_subs(descs, counts, nsub, kWgV3KS, subs);
      PlanOut po;
      wgrad_plan(subs, nsub, cfg0, wgrad_multi_cap(descs, counts, nsub), &po);
      for (int s = 0; s < nsub; ++s) splits[s] = po.splits[s];
      return 0;
    }
  }
  long long total = 0;
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    total += (long long)ktiles * tiles * counts[s];
    off += counts[s];
  }
  const int cfg = wgrad_pick(descs);
  const int slots = wgrad_multi_cap(descs, counts, nsub) * (cfg == 4 ? 2 : 1);
  long long lmax = (total + slots - 1) / slots;
  if (lmax < 4) lmax = 4;
  off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    int sp = (int)((ktiles + lmax - 1) / lmax);
    const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;
    if (sp > max_by_k) sp = max_by_k;
    if (sp < 1) sp = 1;
    splits[s] = sp;
    off += counts[s];
  }
  return 0;
}

int wgrad_multi_check(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  DSL_CHECK(descs && counts && nsub >= 1 && nsub <= kMaxMulti, "dsl_wgrad_multi: bad sub-launch list (n=%d)", nsub);
  const int cfg = wgrad_pick(descs);
  DSL_CHECK(cfg >= 1 && cfg <= 4, "dsl_wgrad_multi: tile configuration %d has no multi form", cfg);
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    DSL_CHECK(counts[s] >= 1 && counts[s] <= DSL_MAX_GROUP, "dsl_wgrad_multi: sub-launch %d has %d members", s, counts[s]);
    const dsl_wgrad_desc* d = &descs[off];
    DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG && d->cs % 128 == 0 && d->cy % 64 == 0 && d->cd <= d->cy,
              "dsl_wgrad_multi: bad geometry in sub-launch %d", s);
    DSL_CHECK(wgrad_pick(d) == cfg, "dsl_wgrad_multi: sub-launch %d needs tile configuration %d, the launch uses %d", s, wgrad_pick(d), cfg);
    for (int g = 0; g < counts[s]; ++g) {
      DSL_CHECK(d[g].dy && d[g].x && d[g].dw, "dsl_wgrad_multi: null pointer (sub-launch %d member %d)", s, g);
      DSL_CHECK(!d[g].shared, "dsl_wgrad_multi: shared-weight groups use dsl_conv2d_wgrad_group");
      DSL_CHECK(wgrad_same_geometry(d, &d[g]), "dsl_wgrad_multi: sub-launch %d member %d has a different geometry", s, g);
    }
    off += counts[s];
  }
  return 0;
}
}  // namespace

extern "C" int dsl_wgrad_multi_config(const dsl_wgrad_desc* d) {
  DSL_CHECK(d != nullptr, "dsl_wgrad_multi_config: null descriptor");
  return d->shared ? 0 : wgrad_pick(d);
}

extern "C" size_t dsl_wgrad_multi_table_bytes(void) { return sizeof(WgMultiTable); }

extern "C" size_t dsl_wgrad_multi_workspace_bytes(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  if (wgrad_multi_check(descs, counts, nsub)) return 0;
  int splits[kMaxMulti];
  wgrad_multi_splits(descs, counts, nsub, splits);
  size_t need = 0;
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    const dsl_wgrad_desc* d = &descs[off];
    if (splits[s] > 1) need += (size_t)splits[s] * counts[s] * wgrad_cy_pad(d) * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);
    off += counts[s];
  }
  return need ? need : 16;
}

// Fills `table_host` (dsl_wgrad_multi_table_bytes()) for the sub-launches descs[0 .. sum(counts)) (sub-launch s = counts[s]
// consecutive same-geometry descriptors, all of one tile configuration, dsl_wgrad_multi_config).  The caller copies the
// bytes to device memory once and passes both copies to dsl_conv2d_wgrad_multi; the table stays valid while the
// descriptors' pointers and `workspace` do.
extern "C" int dsl_wgrad_multi_build(const dsl_wgrad_desc* descs, const int* counts, int nsub, void* workspace, size_t ws_bytes,
                                     void* table_host, size_t table_bytes) {
  if (int rc = wgrad_multi_check(descs, counts, nsub)) return rc;
  DSL_CHECK(table_host && table_bytes >= sizeof(WgMultiTable), "dsl_wgrad_multi_build: table buffer too small");
  DSL_CHECK(workspace && ws_bytes >= dsl_wgrad_multi_workspace_bytes(descs, counts, nsub), "dsl_wgrad_multi_build: workspace too small");
  WgMultiTable* t = (WgMultiTable*)table_host;
  memset(t, 0, sizeof(*t));
  t->magic = kMultiMagic;
  t->cfg = wgrad_pick(descs);
  t->nsub = nsub;
  int splits[kMaxMulti], first[kMaxMulti], order[kMaxMulti];
  wgrad_multi_splits(descs, counts, nsub, splits);
  {                                    // the pipelined kernel serves the launch only if it can serve every sub-launch
    bool v3 = true;
    int off = 0;
    for (int s = 0; s < nsub; ++s) { v3 = v3 && wgrad_v3_ok(&descs[off], t->cfg); off += counts[s]; }
    t->v3 = v3 ? 1 : 0;
  }
  long long per_wg[kMaxMulti];
  {
    int off = 0;
    for (int s = 0; s < nsub; ++s) {
      int ktiles, tiles, bco;
      wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
      per_wg[s] = (ktiles + splits[s] - 1) / splits[s];
      first[s] = off;
      order[s] = s;
      off += counts[s];
    }
  }
  for (int i = 1; i < nsub; ++i)            // longest workgroups first (stable insertion sort)
    for (int j = i; j > 0 && per_wg[order[j]] > per_wg[order[j - 1]]; --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
  unsigned char* ws = (unsigned char*)workspace;
  int blocks = 0, red_blocks = 0;
  for (int i = 0; i < nsub; ++i) {
    const int s = order[i];
    const dsl_wgrad_desc* d = &descs[first[s]];
    WgK& k = t->k[i];
    long long px, xo;
    if (int rc = wgrad_fill_k(d, counts[s], splits[s], t->cfg, t->v3 != 0, k, &px, &xo)) return rc;
    k.direct = splits[s] == 1 ? 1 : 0;
    k.ws = (float*)ws;
    blocks += k.chunk * 8;
    t->hdr.wg_end[i] = blocks;
    t->flops += 2.0 * counts[s] * px * (double)d->cd * d->kh * d->kw * d->cs;
    t->bytes += counts[s] * ((double)px * d->cd * 2.0 + (double)xo * d->cs * 2.0 + (double)d->cd * d->kh * d->kw * d->cs * 4.0);
    const long long sub_elems = (long long)counts[s] * k.cyp * k.krow;
    k.dbws = (float*)ws + (size_t)splits[s] * sub_elems;       // behind this sub-launch's dW partials
    for (int g = 0; g < counts[s]; ++g) {
      if (!k.direct) {
        DSL_CHECK(t->n_red < kMaxRed, "dsl_wgrad_multi_build: more than %d reduce entries", kMaxRed);
        RedEnt& r = t->red[t->n_red++];
        r.ws = (const float*)ws + (long long)g * k.cyp * k.krow;
        r.dw = d[g].dw; r.scale = d[g].scale; r.db = d[g].db;
        r.dbws = k.dbws + (long long)g * k.cyp; r.dbstride = (long long)counts[s] * k.cyp;
        r.krow = k.krow; r.sstride = sub_elems; r.splits = splits[s]; r.cd = d->cd;
        const long long total4 = (long long)d->cd * k.krow / 4;
        int nb = (int)((total4 + 1023) / 1024);        // ~4 f32x4 per thread
        if (nb > 512) nb = 512;
        if (nb < 1) nb = 1;
        r.blk_start = red_blocks; r.nblk = nb;
        red_blocks += nb;
      }
    }
    if (!k.direct) ws += (size_t)splits[s] * (sub_elems + (long long)counts[s] * k.cyp) * sizeof(float);
  }
  t->hdr.nsub = nsub;
  t->total_blocks = blocks;
  t->red_blocks = red_blocks;
  if (wgrad_plan_mode() && t->v3 && t->cfg <= 3 && wgrad_persist()) {
    // the schedule of the persistent grid, for the sub-launches in TABLE order (that is the block numbering the kernel sees)
    PlanSub subs[kMaxMulti], tsubs[kMaxMulti];
    int tsplits[kMaxMulti];
    wgrad_plan_subs(descs, counts, nsub, kWgV3KS, subs);
    long long items = 0;
    for (int i = 0; i < nsub; ++i) { tsubs[i] = subs[order[i]]; tsplits[i] = splits[order[i]]; items += (long long)tsplits[i] * tsubs[i].tiles; }
    const int cap_ = wgrad_multi_cap(descs, counts, nsub);
    int G = (int)std::min<long long>(cap_, (items + 7) / 8 * 8);
    if (G < 8) G = 8;
    {
      const long long ms_cap = plan_simulate(tsubs, tsplits, nsub, cap_, plan_ovh(t->cfg), nullptr, 0, nullptr, nullptr);
      for (; G < cap_; G += 8)
        if (plan_simulate(tsubs, tsplits, nsub, G, plan_ovh(t->cfg), nullptr, 0, nullptr, nullptr) <= ms_cap) break;
    }
    int rounds = 0, n_items = 0;
    const long long ms = plan_simulate(tsubs, tsplits, nsub, G, plan_ovh(t->cfg), t->sched, kSchedMax, &rounds, &n_items);
    if (blocks < 32767 && rounds > 0) { t->sched_grid = G; t->sched_rounds = rounds; }
    t->plan_makespan = (int)ms; t->plan_items = n_items;
  }
  return 0;
}

// Planner probe (tests, tools; no device needed): sub-launch s has stages[s] K stages, tiles[s] output tiles (all members) and
// tile_elems[s] elements per tile set; returns the chosen split factors and info = {grid, makespan, items, makespan of round
// 3's rule under the stride walk, its items}
extern "C" int dsl_wgrad_plan_probe(const int* stages, const int* tiles, const long long* tile_elems, int nsub, int cfg, int cap,
                                    int* splits_out, int* info) {
  DSL_CHECK(stages && tiles && nsub >= 1 && nsub <= kMaxMulti && cfg >= 1 && cfg <= 3 && cap >= 8 && cap % 8 == 0, "dsl_wgrad_plan_probe: bad arguments");
  PlanSub subs[kMaxMulti];
  for (int s = 0; s < nsub; ++s) {
    subs[s].stages = stages[s]; subs[s].tiles = tiles[s]; subs[s].max_sp = std::max(1, stages[s] / 8);
    subs[s].tile_elems = tile_elems ? tile_elems[s] : 0;
  }
  PlanOut po;
  wgrad_plan(subs, nsub, cfg, cap, &po);
  for (int s = 0; s < nsub; ++s) splits_out[s] = po.splits[s];
  info[0] = po.grid; info[1] = po.makespan; info[2] = po.items;
  {   // self-check of the schedule table the launch would use: every valid block exactly once, no holes in a workgroup's list
    std::vector<short> sched(kSchedMax);
    int rounds = 0, n_items = 0, total_blocks = 0;
    plan_simulate(subs, po.splits, nsub, po.grid, plan_ovh(cfg), sched.data(), kSchedMax, &rounds, &n_items);
    for (int s = 0; s < nsub; ++s) total_blocks += (subs[s].tiles * po.splits[s] + 7) / 8 * 8;
    if (rounds > 0) {
      std::vector<int> hit(total_blocks, 0);
      int seen_items = 0;
      for (int b = 0; b < po.grid; ++b) {
        bool ended = false;
        for (int r = 0; r < rounds; ++r) {
          const int vb = sched[r * po.grid + b];
          if (vb < 0) { ended = true; continue; }
          DSL_CHECK(!ended && vb < total_blocks && (vb & 7) == (b & 7) && hit[vb]++ == 0, "dsl_wgrad_plan_probe: bad schedule entry (workgroup %d round %d block %d)", b, r, vb);
          ++seen_items;
        }
      }
      DSL_CHECK(seen_items == n_items && n_items == po.items, "dsl_wgrad_plan_probe: schedule holds %d of %d items", seen_items, n_items);
    }
  }
  // round 3: one target length, stride walk of the blocks in "longest per workgroup first" order
  long long total = 0;
  for (int s = 0; s < nsub; ++s) total += (long long)((stages[s] + 1) / 2) * tiles[s];
  long long lmax = std::max<long long>(4, (total + cap - 1) / cap);
  int osp[kMaxMulti], ord[kMaxMulti];
  for (int s = 0; s < nsub; ++s) {
    const int kt = (stages[s] + 1) / 2;
    int sp = (int)((kt + lmax - 1) / lmax);
    sp = std::max(1, std::min(sp, std::max(1, kt / 4)));
    osp[s] = sp; ord[s] = s;
  }
  for (int i = 1; i < nsub; ++i)
    for (int j = i; j > 0 && (stages[ord[j]] + osp[ord[j]] - 1) / osp[ord[j]] > (stages[ord[j - 1]] + osp[ord[j - 1]] - 1) / osp[ord[j - 1]]; --j) std::swap(ord[j], ord[j - 1]);
  std::vector<long long> load(cap, 0);
  int base = 0, oitems = 0;
  for (int i = 0; i < nsub; ++i) {
    const int s = ord[i], witems = tiles[s] * osp[s], chunk = (witems + 7) / 8, tps = (stages[s] + osp[s] - 1) / osp[s];
    for (int bid = 0; bid < chunk * 8; ++bid) {
      const int xcd = bid & 7, jj = bid >> 3, w = xcd * chunk + jj;
      if (jj >= chunk || w >= witems) continue;
      const int sp = w / tiles[s], k0 = sp * tps, k1 = std::min(k0 + tps, stages[s]);
      if (k1 > k0) { load[(base + bid) % cap] += plan_ovh(cfg) + k1 - k0; ++oitems; }
    }
    base += chunk * 8;
  }
  info[3] = (int)*std::max_element(load.begin(), load.end());
  info[4] = oitems;
  return 0;
}

// what a table holds (profiling tools): algorithmic flops / bytes of the launch, its workgroups, reduce workgroups, sub-launches
extern "C" int dsl_wgrad_multi_info(const void* table_host, double* flops, double* bytes, int* blocks, int* red_blocks, int* nsub) {
  const WgMultiTable* t = (const WgMultiTable*)table_host;
  DSL_CHECK(t && t->magic == kMultiMagic, "dsl_wgrad_multi_info: not a table of dsl_wgrad_multi_build");
  if (flops) *flops = t->flops;
  if (bytes) *bytes = t->bytes;
  if (blocks) *blocks = t->total_blocks;
  if (red_blocks) *red_blocks = t->red_blocks;
  if (nsub) *nsub = t->nsub;
  return 0;
}

extern "C" int dsl_conv2d_wgrad_multi(const void* table_host, const void* table_dev, void* stream) {
  const WgMultiTable* t = (const WgMultiTable*)table_host;
  DSL_CHECK(t && table_dev && t->magic == kMultiMagic, "dsl_conv2d_wgrad_multi: not a table of dsl_wgrad_multi_build");
  const WgK* ktab = (const WgK*)((const unsigned char*)table_dev + offsetof(WgMultiTable, k));
  const RedEnt* rtab = (const RedEnt*)((const unsigned char*)table_dev + offsetof(WgMultiTable, red));
  hipStream_t st = (hipStream_t)stream;
  const int prof = dsl_prof_active() ? dsl_prof_begin(3, t->flops, st, t->bytes) : -1;
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128}, nsts[5] = {2, 2, 3, 3, 2};
  const size_t lds2 = t->v3 ? wgrad_v3_lds(t->cfg) : (size_t)nsts[t->cfg] * 64 * 2 * (bcos[t->cfg] + bcis[t->cfg]);
  const dim3 grid(t->total_blocks);
#define LAUNCHM(KERNEL, A, B, C_, D, KS_, S_)                                                                        \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)KERNEL<A, B, C_, D, KS_, S_>,                                                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((KERNEL<A, B, C_, D, KS_, S_>), grid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab);              \
  } while (0)
  const int cap = (wgrad_slots() + 7) / 8 * 8;
  if (t->v3 && t->sched_rounds > 0) {
    const short* sched = (const short*)((const unsigned char*)table_dev + offsetof(WgMultiTable, sched));
    const dim3 sgrid(t->sched_grid);
#define LAUNCHS(A, B, C_, D, KS_, S_)                                                                                \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)wgrad_pipe_multi_sched_kernel<A, B, C_, D, KS_, S_>,                          \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((wgrad_pipe_multi_sched_kernel<A, B, C_, D, KS_, S_>), sgrid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab,   \
                       sched, t->sched_rounds);                                                                      \
  } while (0)
    switch (t->cfg) {
      case 1: LAUNCHS(256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHS(256, 128, 4, 2, 32, 5); break;
      default: LAUNCHS(128, 256, 2, 4, 32, 5); break;
    }
#undef LAUNCHS
  } else if (t->v3 && wgrad_persist() && t->total_blocks > cap) {
    const dim3 pgrid(cap);
#define LAUNCHP(A, B, C_, D, KS_, S_)                                                                                \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)wgrad_pipe_multi_persist_kernel<A, B, C_, D, KS_, S_>,                        \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((wgrad_pipe_multi_persist_kernel<A, B, C_, D, KS_, S_>), pgrid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab, \
                       t->total_blocks);                                                                             \
  } while (0)
    switch (t->cfg) {
      case 1: LAUNCHP(256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHP(256, 128, 4, 2, 32, 5); break;
      default: LAUNCHP(128, 256, 2, 4, 32, 5); break;
    }
#undef LAUNCHP
  } else if (t->v3) {
    switch (t->cfg) {
      case 1: LAUNCHM(wgrad_pipe_multi_kernel, 256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHM(wgrad_pipe_multi_kernel, 256, 128, 4, 2, 32, 5); break;
      default: LAUNCHM(wgrad_pipe_multi_kernel, 128, 256, 2, 4, 32, 5); break;
    }
  } else {
    switch (t->cfg) {
      case 1: LAUNCHM(wgrad_glds_multi_kernel, 256, 256, 2, 4, 64, 2); break;
      case 2: LAUNCHM(wgrad_glds_multi_kernel, 256, 128, 4, 2, 64, 3); break;
      case 3: LAUNCHM(wgrad_glds_multi_kernel, 128, 256, 2, 4, 64, 3); break;
      default: LAUNCHM(wgrad_glds_multi_kernel, 128, 128, 2, 2, 64, 2); break;
    }
  }
#undef LAUNCHM
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("wgrad_glds_multi_kernel");
  if (t->n_red > 0) {
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(t->red_blocks), dim3(256), 0, st, rtab, t->n_red);
    DSL_LAUNCH_CHECK("wgrad_reduce_multi_kernel");
  }
  for (int i = 0; i < t->n_colsum; ++i) {
    const ColsumItem& c = t->colsum[i];
    const int rc = c.clear ? dsl_colsum(c.x, c.out, (long)c.rows, c.c, c.ld, stream) : dsl_colsum_acc(c.x, c.out, (long)c.rows, c.c, c.ld, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int dsl_conv2d_wgrad(const dsl_wgrad_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv2d_wgrad: null descriptor");
  return wgrad_launch(d, 1, stream);
}

extern "C" int dsl_conv2d_wgrad_group(const dsl_wgrad_desc* descs, int count, void* stream) {
  DSL_CHECK(descs != nullptr, "dsl_conv2d_wgrad_group: null descriptors");
  return wgrad_launch(descs, count, stream);
}
