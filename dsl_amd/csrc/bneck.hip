// A whole TRAINED bottleneck's forward pass as ONE kernel (round 5):
//   a1  = relu(bn1(conv1_1x1[/stride](x)))          -> written (conv2's weight gradient reads it)
//   a2  = relu(bn2(conv2_3x3(a1)))                  -> written (conv3's weight gradient reads it)
//   out = relu(bn3(conv3_1x1(a2)) + identity)
// (reference: mmdet/models/backbones/resnet.py:262-301 `Bottleneck.forward`, caffe style: the stride sits on conv1, :153-158;
// eval-mode BatchNorms folded to per-channel (scale, bias), :647-656; identity = x or the downsample branch's output, :284-286.)
//
// Why (profiles/r05_step_ablation_regions.txt): with the forward convolutions of layer2 / layer3 / layer4 left out the
// step is 0.39 / 0.40 / 0.22 ms shorter (1.01 ms of 4.60 for the three together, 11 % of the step's arithmetic) while the same
// stages' data gradients cost 0.30 ms - the forward chains run alone on the chip, two half-batch chains of 16 - 30 us launches of
// which ~8 us each is launch, fill and epilogue, and layer2's 1x1 convolutions move the 512-channel tensors through HBM three
// times per block.  Here a workgroup owns a TY x TX pixel tile of ONE image for the whole block: conv1 is computed for the tile plus
// a one-pixel halo straight out of global memory (x by DMA, K tile by K tile) into an LDS patch P1 (zero outside the image: conv2's
// padding), conv2 multiplies its nine taps out of P1 by address arithmetic, conv3 reads the staged conv2 tile P2; every weight
// streams through a two-stage DMA ring exactly once per workgroup.  x is read once (+ the halo, an L2 hit) and once more as the
// identity, a1 / a2 / out are written once; one launch instead of three (six: the image-split chains).
//
// Arithmetic = the three dsl_conv2d launches it replaces, operation for operation: same K order per MFMA chain (channel blocks
// ascending for the 1x1s; tap-major, then channel blocks for the 3x3 - conv_pipe_kernel's order), a1 / a2 rounded to bf16 where the
// separate launches store them, epilogues mul, add, (+ identity), ReLU, one rounding each (fp contract off).
// tests/test_kernels_gpu.py::test_bottleneck_forward_fused compares bits.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#pragma clang fp contract(off)
__device__ __forceinline__ float mul_nc(float a, float b) { return a * b; }
__device__ __forceinline__ float add_nc(float a, float b) { return a + b; }
#pragma clang fp contract(fast)

struct BnK {
  const uint16_t* x; const uint16_t* w1; const uint16_t* w2; const uint16_t* w3; const uint16_t* idt;
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  uint16_t* a1; uint16_t* a2; uint16_t* out;
  int n, hin, win, h, w;          // source / output spatial sizes (h = hin / stride ...)
  int cin, ldx, stride;           // conv1 reads `cin` channels of rows of ldx elements at pixel stride `stride`
  int ldi, ldo;                   // row strides of identity / out (elements)
  int tiles_y, tiles_x, ntiles, xcd_chunk;
  int dbg;                        // timing probe of the ablation build (tools/build_ablate.sh, DSL_BNECK_DBG; results are wrong): 1 = stop behind phase A, 2 = behind phase B
};

// P: planes (128 = layer2, 256 = layer3); TY x TX: output pixels per workgroup.  512 threads = 8 waves: 4 cout groups x 2 pixel groups.
//
// LDS map (160 KB; NS = 3 ring stages everywhere: two K tiles in flight beside the one being multiplied - with one in flight the
// profile showed 51 - 56 % of the wave cycles parked at the tile-top wait, profiles/r05_bneck_pmc_v2.txt):
//   phase A      ring of 3 stages [weights P x 128 B | x HPR x 128 B] from offset 0;  P1 (the a1 patch, [HPR][P] bf16) is written at
//                offset 96 K only in the phase's epilogue, over the then dead third stage
//   phases B, C  ONE weight ring of 3 stages [P x 128 B] from offset 0 that runs through conv2's 9 P / 64 tiles and on through
//                conv3's 4 x P / 64 tiles without a bubble;  P2 (the a2 tile, [TPR][P]) between the ring and P1 (layer2) or over the
//                dead P1 (layer3), written in phase B's epilogue;  S (identity / output staging of a cout block) over the dead P1
template <int P, int TY, int TX>
struct Cfg {
  static constexpr int T = 512, WCO = 4, WPX = 2, NS = 3;
  static constexpr int C = 4 * P;
  static constexpr int CT = P / (32 * WCO);                  // 32-cout MFMA tiles per wave (1 or 2)
  static constexpr int TP = TY * TX;                         // tile pixels
  static constexpr int PTB = (TP + 63) / 64;                 // 32-pixel MFMA tiles per wave, phases B / C
  static constexpr int TPR = PTB * 64;                       // tile rows incl. padding lanes
  static constexpr int PW = TX + 2;                          // patch width
  static constexpr int HP = (TY + 2) * PW;                   // halo pixels
  static constexpr int PTA = (HP + 63) / 64;                 // ... phase A
  static constexpr int HPR = PTA * 64;
  static constexpr int NCH = P / 8;                          // 16-byte chunks per staged row
  static constexpr int ROW = P * 2;                          // bytes per staged row (P1, P2, S)
  static constexpr int WST = P * 128;                        // bytes of one weight stage: [P rows][64 K]
  static constexpr int XST = HPR * 128;                      // ... of one x stage
  static constexpr int AST = WST + XST;                      // phase A stage
  static constexpr int OFF_P1 = 96 * 1024;
  static constexpr int OFF_P2 = (NS * WST + TPR * ROW <= OFF_P1) ? NS * WST : OFF_P1;
  static constexpr int OFF_S = (OFF_P2 == OFF_P1) ? OFF_P1 + TPR * ROW : OFF_P1;
  static constexpr int LDS = 160 * 1024;
  static_assert(NS * AST <= LDS && OFF_P1 + HPR * ROW <= LDS, "phase A ring and P1");
  static_assert(NS * WST <= OFF_P2 || OFF_P2 == OFF_P1, "weight ring below P2");
  static_assert(NS * WST <= OFF_P1 && OFF_S + TPR * ROW <= LDS && OFF_P2 + TPR * ROW <= (OFF_P2 == OFF_P1 ? OFF_S : OFF_P1), "P2 / S placement");
  static_assert(P % 64 == 0 && (TX == 16 || TX == 12), "patch swizzles are fitted to these widths");
  // conflict-free patch swizzle for the 3x3's shifted 16-byte fragment reads (every ds_read_b128 lane group meets 16 distinct bank
  // quads for all nine taps): slot = chunk ^ f(patch row R, patch column cx)
  __device__ static __forceinline__ int f1(int R, int cx) { return TX == 16 ? (cx & 15) : ((4 * R + 3 * cx) & 15); }
};

// NMF x { 1 MFMA [, 1 DS read for the first NDS] [, 1 VMEM for the first NVM] }: the fragment reads of K step kk + 1 and the DMA pieces
// of the tile being fetched go out between the MFMAs of step kk
template <int NMF, int NDS, int NVM>
__device__ __forceinline__ void sched_mix() {
#pragma unroll
  for (int i = 0; i < NMF; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (i < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
}

// s_waitcnt vmcnt(n) for the handful of counts the tile loops need (an immediate operand)
template <int A, int B, int C_>
__device__ __forceinline__ void wait_vm(int n) {
  if (n >= A + B + C_) wait_vmcnt<A + B + C_>();
  else if (n >= A + B) wait_vmcnt<A + B>();
  else if (n >= A) wait_vmcnt<A>();
  else wait_vmcnt<0>();
}

template <int P, int TY, int TX>
__device__ __forceinline__ void bneck_fwd_body(const BnK& p, unsigned char* smem) {
  using K = Cfg<P, TY, TX>;
  constexpr int CT = K::CT, PTA = K::PTA, PTB = K::PTB, PW = K::PW, ROW = K::ROW, NCH = K::NCH, T = K::T, NS = K::NS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave >> 1, wave_px = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  // XCD-aware tile order (as conv_pipe_kernel): workgroups with equal b % 8 share an XCD and own a contiguous run of tiles
  const int wi = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.xcd_chunk || wi >= p.ntiles) return;
  const int img = wi / (p.tiles_y * p.tiles_x);
  const int trem = wi - img * (p.tiles_y * p.tiles_x);
  const int ty0 = (trem / p.tiles_x) * TY, tx0 = (trem % p.tiles_x) * TX;

  unsigned char* const P1 = smem + K::OFF_P1;
  unsigned char* const P2 = smem + K::OFF_P2;
  unsigned char* const S = smem + K::OFF_S;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_id = __builtin_amdgcn_make_buffer_rsrc((void*)p.idt, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a2, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);

  // Static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH: the younger wave of a SIMD otherwise loses every
  // issue arbitration and reaches each barrier late)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

  // ---- DMA lanes: thread -> (row within a 64-row pass, 16-byte slot); the source chunk carries the ring's XOR swizzle
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
  constexpr int WPASS = P / 64, XPASS = K::HPR / 64;
  // one 64-row piece of a weight stage: rows [row0 + i * 64, + 64) of a [rows][wrow] bf16 matrix, K tile kt -> LDS at `dst`
  auto dma_w_piece = [&](const __amdgpu_buffer_rsrc_t& rs, int wrow_bytes, int row0, int kt, unsigned char* dst, int i) {
    const unsigned v = (unsigned)((row0 + i * 64 + lrow) * wrow_bytes + chunk * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + (i * 64 + wave * 8) * 128), 16, v, (unsigned)(kt * 128), 0, 0);
  };
  // x stage: halo pixel j = pass * 64 + lrow -> source pixel (out of the image / beyond the patch: zeros)
  unsigned xoff[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int j = i * 64 + lrow;
    const int hy = j / PW, hx = j - hy * PW;
    const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
    const bool ok = j < K::HP && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
    const int pix = (img * p.hin + y * p.stride) * p.win + x * p.stride;
    xoff[i] = ok ? (unsigned)(pix * p.ldx + chunk * 8) * 2u : 0x80000000u;
  }
  auto dma_x_piece = [&](int kt, unsigned char* dst, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(dst + (i * 64 + wave * 8) * 128), 16, xoff[i], (unsigned)(kt * 128), 0, 0);
  };
  // The tile's own pixels as 16-byte store items: item id -> (tile pixel t, chunk c); every thread issues the SAME number of store
  // instructions (out-of-range offsets drop the surplus), so the vmcnt bookkeeping around the stores is a compile-time constant
  constexpr int NSTO = (K::TP * NCH + T - 1) / T;
  // (item -> pixel decoded where it is used: a handful of integer operations per item and phase instead of 3 x NSTO registers held
  // through the K loops)
  auto sto_item = [&](int i, int& t, int& c, unsigned& pix) {
    const int id = tid + i * T;
    t = id / NCH;
    c = id - t * NCH;
    const int py = t / TX, px = t - py * TX;
    const int y = ty0 + py, x = tx0 + px;
    const bool ok = id < K::TP * NCH && y < p.h && x < p.w;
    pix = ok ? (unsigned)((img * p.h + y) * p.w + x) : 0x80000000u;
    if (!ok) t = 0;
  };
  u32x4 sto_r[NSTO];                               // staged rows on their way to memory: read from LDS at the end of a phase, stored at the
                                                   // next sync point (a clean point of the vmcnt ledger, see the tile loops)
  auto store_rows = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned ld, unsigned col0) {
#pragma unroll
    for (int i = 0; i < NSTO; ++i) {
      int t, c;
      unsigned pix;
      sto_item(i, t, c, pix);
      const unsigned v = pix == 0x80000000u ? 0x80000000u : (pix * ld + col0 + (unsigned)c * 8u) * 2u;
      __builtin_amdgcn_raw_buffer_store_b128(sto_r[i], rs, v, 0, 0);
    }
  };

  // ---- fragment helpers (two register sets: the reads of K step kk + 1 fly during the MFMAs of step kk)
  bf16x8 fa[4][CT];                                // one fragment set per K step of a tile: the reads run HALF A TILE ahead of the MFMAs
  auto read_a = [&](const unsigned char* wst, int kk, int f) {
    const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[f][ct] = *reinterpret_cast<const bf16x8*>(wst + (wave_co * (32 * CT) + ct * 32 + frow) * 128 + coff);
  };

  // =================================================================================================================
  // Phase A: a1 on the halo patch = relu(bn1(W1 . x)); GEMM [P couts] x [HPR halo pixels] x [cin]
  // =================================================================================================================
  {
    f32x16 acc[CT][PTA];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int b = 0; b < PTA; ++b)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
    const int KT = p.cin >> 6;
    const int wrow = p.cin * 2;
    constexpr int LPT = WPASS + XPASS;
    bf16x8 fb[4][PTA];
    auto piece = [&](int kt, int i) {               // piece i of tile kt's stage (weights first)
      unsigned char* st = smem + (kt % NS) * K::AST;
      if (i < WPASS) dma_w_piece(rs_w1, wrow, 0, kt, st, i);
      else dma_x_piece(kt, st + K::WST, i - WPASS);
    };
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < KT) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) piece(s_, i);
      }
    auto rd = [&](int kt, int kk, int f) {
      const unsigned char* st = smem + (kt % NS) * K::AST;
      read_a(st, kk, f);
      const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
#pragma unroll
      for (int pt = 0; pt < PTA; ++pt)
        fb[f][pt] = *reinterpret_cast<const bf16x8*>(st + K::WST + ((wave_px * PTA + pt) * 32 + frow) * 128 + coff);
    };
    auto mma = [&](int f) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PTA; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f][ct], fb[f][pt], acc[ct][pt], 0, 0, 0);
    };
    // Software pipeline (conv_pipe_kernel's): the tile's single barrier sits in front of its LAST K step's MFMAs, whose fragments are in
    // registers by then; behind it the next tile has landed for every wave, its first fragments are requested at once and fly while
    // those MFMAs run - the matrix pipe does not drain at the tile boundary.
    // (a K step of this kernel is only CT x PT = 3 - 4 MFMAs per wave: with the reads ONE step ahead every step waited for its LDS
    // latency - 1 700 - 1 900 cycles per tile against 770 - 1 020 of MFMA, profiles/r05_bneck_micro_v4.txt; half a tile ahead they are
    // covered by 6 - 8 MFMAs of the wave and as many of its SIMD partner)
    wait_vm<LPT, 0, 0>(min(NS - 2, KT - 1) * LPT);
    lds_barrier();
    rd(0, 0, 0);
    rd(0, 1, 1);
    for (int kt = 0; kt < KT; ++kt) {
      const bool more = kt + NS - 1 < KT;
      rd(kt, 2, 2);
      rd(kt, 3, 3);
      if (more) {                                  // the fetched tile's DMA pieces: all out before the wait below
#pragma unroll
        for (int i = 0; i < LPT; ++i) piece(kt + NS - 1, i);
      }
      mma(0);
      mma(1);
      sched_mix<2 * CT * PTA, 2 * (CT + PTA), LPT>();
      if (kt + 1 < KT) {
        wait_vm<LPT, 0, 0>(min(NS - 2, KT - 2 - kt) * LPT);      // tile kt + 1 has landed (this thread's pieces) ...
        lds_barrier();                                           // ... everyone's; and everyone holds tile kt's last fragments: its stage is free
        rd(kt + 1, 0, 0);
        rd(kt + 1, 1, 1);
      }
      mma(2);
      mma(3);
      sched_mix<2 * CT * PTA, 2 * (CT + PTA), 0>();
    }
    wait_vmcnt<0>();
    lds_barrier();                                 // every wave is through its last fragments: the ring is free, P1 may be written
    // conv2's first two weight tiles fly during the epilogue
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
#pragma unroll
      for (int i = 0; i < WPASS; ++i) dma_w_piece(rs_w2, 9 * P * 2, 0, s_, smem + s_ * K::WST, i);
    // epilogue A: scale, bias, ReLU, round -> P1[halo row][cout] (zero outside the image: conv2's padding)
    // (cout group outermost: a lane keeps ONE group's scale / bias at a time - all of them at once cost 64 registers at P = 256)
    int jrow[PTA], jsw[PTA];
    bool jok[PTA];
#pragma unroll
    for (int pt = 0; pt < PTA; ++pt) {
      const int j = (wave_px * PTA + pt) * 32 + frow;
      const int hy = j / PW, hx = j - hy * PW;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      jrow[pt] = j;
      jok[pt] = j < K::HP && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
      jsw[pt] = K::f1(hy, hx);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s1 + col);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b1 + col);
#pragma unroll
        for (int pt = 0; pt < PTA; ++pt) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(add_nc(mul_nc(acc[ct][pt][4 * g + e], sc[e]), bi[e]), 0.f);
            if (!jok[pt]) v[e] = 0.f;
          }
          const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(P1 + jrow[pt] * ROW + ((((col >> 3) ^ jsw[pt]) << 4) | (fhalf << 3))) = o;
        }
      }
    lds_barrier();
    // a1 (the tile's own pixels): rows into registers now, to memory at phase B's first tile top
#pragma unroll
    for (int i = 0; i < NSTO; ++i) {
      int t, c;
      unsigned pix;
      sto_item(i, t, c, pix);
      const int py = t / TX, px = t - py * TX;
      sto_r[i] = *reinterpret_cast<const u32x4*>(P1 + ((py + 1) * PW + px + 1) * ROW + ((c ^ K::f1(py + 1, px + 1)) << 4));
    }
  }
#ifdef DSL_ABLATE_BUILD
  if (p.dbg == 1) { wait_vmcnt<0>(); return; }
#endif

  // =================================================================================================================
  // Phases B and C share ONE weight-tile stream: q = 0 .. QB - 1 conv2's tiles (tap-major, then channel blocks), then conv3's
  // (cout block, K tile).  Tile q sits in ring stage q % NS; at tile q's top tile q + NS - 1 is issued (one 64-row piece per K step).
  // vmcnt ledger: besides the weight tiles a thread issues its row stores (NSTO) and the identity DMA (SDMA) - always at a tile top,
  // right behind the wait and the barrier, in front of the fetched tile's pieces: the NEXT tile top then waits for a tile that is
  // older than them and allows them to be outstanding (+ X), every later one waits for a tile that is younger (they are done).
  // =================================================================================================================
  constexpr int KCB = P / 64;                      // K tiles per tap (conv2) and per cout block (conv3)
  constexpr int QB = 9 * KCB, QC = 4 * KCB, QT = QB + QC;
  constexpr int SDMA = K::TPR * ROW / (T * 16);    // identity DMA instructions per wave and cout block
  static_assert((K::TPR * ROW) % (T * 16) == 0, "whole DMA instructions per wave");
  static_assert(WPASS <= 4, "one weight piece per K step");
  auto wpiece = [&](int q, int i) {
    unsigned char* st = smem + (q % NS) * K::WST;
    if (q < QB) dma_w_piece(rs_w2, 9 * P * 2, 0, q, st, i);
    else {
      const int qq = q - QB, blk = qq / KCB;
      dma_w_piece(rs_w3, P * 2, blk * P, qq - blk * KCB, st, i);
    }
  };
  int xprev = 0;                                   // non-tile VMEM instructions issued at the previous sync point
  int tpy[PTB], tpx[PTB];                          // this lane's tile pixels (phases B, C): row / column inside the tile
#pragma unroll
  for (int pt = 0; pt < PTB; ++pt) {
    int t = (wave_px * PTB + pt) * 32 + frow;
    if (t >= K::TP) t = 0;                         // padding lanes compute pixel 0 again (never stored)
    tpy[pt] = t / TX;
    tpx[pt] = t - tpy[pt] * TX;
  }
  f32x16 acc[CT][PTB];
  bf16x8 fb[4][PTB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int b = 0; b < PTB; ++b)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
  };
  auto mma = [&](int f) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PTB; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f][ct], fb[f][pt], acc[ct][pt], 0, 0, 0);
  };
  // sync point in front of tile q: tile q has landed for every wave, every wave holds the previous tile's last fragments (its stage is
  // free) and has finished whatever it read from P2 / S; the ledger is clean behind it
  auto sync_for = [&](int q) {
    const int fly = min(NS - 2, QT - 1 - q) * WPASS;
    wait_vm<WPASS, NSTO, SDMA>(fly + xprev);
    lds_barrier();
    xprev = 0;
  };
  unsigned pb[PTB], psw[PTB];                      // conv2: patch row base and swizzle of the lane's pixels for the tap being READ
  auto set_tap = [&](int tap) {
    const int tap_r = tap / 3, tap_s = tap - tap_r * 3;
#pragma unroll
    for (int pt = 0; pt < PTB; ++pt) {
      const int R = tpy[pt] + tap_r, cx = tpx[pt] + tap_s;
      pb[pt] = (unsigned)((R * PW + cx) * ROW);
      psw[pt] = (unsigned)(K::f1(R, cx) << 4);
    }
  };
  auto rd_b = [&](int q, int kk, int f) {          // conv2 tile q: weights from its stage, pixels from P1 at the current tap
    read_a(smem + (q % NS) * K::WST, kk, f);
    const unsigned ch = (unsigned)(((q % KCB) * 8 + 2 * kk + fhalf) << 4);
#pragma unroll
    for (int pt = 0; pt < PTB; ++pt) fb[f][pt] = *reinterpret_cast<const bf16x8*>(P1 + pb[pt] + (ch ^ psw[pt]));
  };
  auto rd_c = [&](int q, int kk, int f) {          // conv3 tile q (K tile k of its cout block): pixels from P2
    read_a(smem + (q % NS) * K::WST, kk, f);
    const int k = (q - QB) % KCB;
#pragma unroll
    for (int pt = 0; pt < PTB; ++pt) {
      const int row = (wave_px * PTB + pt) * 32 + frow;
      fb[f][pt] = *reinterpret_cast<const bf16x8*>(P2 + row * ROW + (((k * 8 + 2 * kk + fhalf) ^ (row & 15)) << 4));
    }
  };
  auto fetch = [&](int q) {                        // DMA pieces of tile q + NS - 1 during tile q's first half: all out before its sync point
    if (q + NS - 1 < QT) {
#pragma unroll
      for (int i = 0; i < WPASS; ++i) wpiece(q + NS - 1, i);
    }
  };
  // ---------------- Phase B: a2 = relu(bn2(W2 * a1)), nine taps out of P1; GEMM [P couts] x [TPR tile pixels] x [9 P]
  zero_acc();
  sync_for(0);
  store_rows(rs_a1, (unsigned)P, 0u);              // a1 leaves for memory (rows read from P1 at the end of phase A)
  xprev = NSTO;
  set_tap(0);
  rd_b(0, 0, 0);
  rd_b(0, 1, 1);
  for (int q = 0; q < QB; ++q) {
    rd_b(q, 2, 2);
    rd_b(q, 3, 3);
    fetch(q);
    mma(0);
    mma(1);
    sched_mix<2 * CT * PTB, 2 * (CT + PTB), WPASS>();
    if (q + 1 < QB) {                              // (the phase's last tile drains: conv3's pixels do not exist yet)
      sync_for(q + 1);
      if ((q + 1) % KCB == 0) set_tap((q + 1) / KCB);
      rd_b(q + 1, 0, 0);
      rd_b(q + 1, 1, 1);
    }
    mma(2);
    mma(3);
    sched_mix<2 * CT * PTB, 2 * (CT + PTB), 0>();
  }
  lds_barrier();                                   // every wave is through P1: P2 (layer3: over P1) may be written
  // epilogue B -> P2[tile pixel][cout] (slot = chunk ^ (row & 15))
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + col);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b2 + col);
#pragma unroll
      for (int pt = 0; pt < PTB; ++pt) {
        const int row = (wave_px * PTB + pt) * 32 + frow;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(add_nc(mul_nc(acc[ct][pt][4 * g + e], sc[e]), bi[e]), 0.f);
        const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2*>(P2 + row * ROW + ((((col >> 3) ^ (row & 15)) << 4) | (fhalf << 3))) = o;
      }
    }
  lds_barrier();
#pragma unroll
  for (int i = 0; i < NSTO; ++i) {
    int t, c;
    unsigned pix;
    sto_item(i, t, c, pix);
    sto_r[i] = *reinterpret_cast<const u32x4*>(P2 + t * ROW + ((c ^ (t & 15)) << 4));
  }
  // ---------------- Phase C: out = relu(bn3(W3 . a2) + identity), four cout blocks of P; GEMM [P couts] x [TPR] x [P] per block
  {
    unsigned idoff[SDMA];                          // this lane's identity source offsets (without the cout block), or 0x80000000
#pragma unroll
    for (int i = 0; i < SDMA; ++i) {
      const int off = (i * 8 + wave) * 1024 + lane * 16;
      const int row = off / ROW, slot_ = (off % ROW) >> 4;
      const int c = slot_ ^ (row & 15);
      const int py = row / TX, px = row - py * TX;
      const int y = ty0 + py, x = tx0 + px;
      const bool ok = row < K::TP && y < p.h && x < p.w;
      idoff[i] = ok ? (unsigned)(((img * p.h + y) * p.w + x) * p.ldi + c * 8) * 2u : 0x80000000u;
    }
    auto dma_idt = [&](int blk) {
#pragma unroll
      for (int i = 0; i < SDMA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_id, (lptr_t)(S + (i * 8 + wave) * 1024), 16, idoff[i], (unsigned)(blk * P * 2), 0, 0);
    };
    // the staged rows of the previous phase / block leave for memory and this block's identity is requested: at a sync point (clean ledger;
    // S is free - every thread's row reads of it are behind the sync's barrier)
    auto block_start = [&](int blk) {
      if (blk == 0) store_rows(rs_a2, (unsigned)P, 0u);
      else store_rows(rs_out, (unsigned)p.ldo, (unsigned)((blk - 1) * P));
      dma_idt(blk);
      xprev = NSTO + SDMA;
    };
    sync_for(QB);
    block_start(0);
    rd_c(QB, 0, 0);
    rd_c(QB, 1, 1);
#ifdef DSL_ABLATE_BUILD
    if (p.dbg == 2) { wait_vmcnt<0>(); return; }
#endif
    for (int blk = 0; blk < 4; ++blk) {
      zero_acc();
      for (int k = 0; k < KCB; ++k) {
        const int q = QB + blk * KCB + k;
        rd_c(q, 2, 2);
        rd_c(q, 3, 3);
        fetch(q);
        mma(0);
        mma(1);
        sched_mix<2 * CT * PTB, 2 * (CT + PTB), WPASS>();
        if (q + 1 < QT) {
          sync_for(q + 1);
          // (a block's identity is requested one tile into the block - behind the previous block's epilogue, which still owns S at the
          // block boundary's own sync point)
          if (k == 0 && blk > 0) block_start(blk);
          rd_c(q + 1, 0, 0);
          rd_c(q + 1, 1, 1);
        }
        mma(2);
        mma(3);
        sched_mix<2 * CT * PTB, 2 * (CT + PTB), 0>();
      }
      // epilogue C of this block: the identity tile must have landed in S.  The sync point behind this block's last tile has already waited
      // for the next tile; what can still be in flight are the NS - 2 tiles after that one - all issued behind the identity request
      // (which went out one tile into the block at the latest): allow exactly those
      {
        const int qlast = QB + blk * KCB + KCB - 1;
        wait_vm<WPASS, 0, 0>(min(NS - 2, QT - 2 - qlast) * WPASS);
      }
      lds_barrier();
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s3 + blk * P + col);
          const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b3 + blk * P + col);
#pragma unroll
          for (int pt = 0; pt < PTB; ++pt) {
            const int row = (wave_px * PTB + pt) * 32 + frow;
            unsigned char* cell = S + row * ROW + ((((col >> 3) ^ (row & 15)) << 4) | (fhalf << 3));
            const u32x2 aa = *reinterpret_cast<const u32x2*>(cell);
            const float ad[4] = {bflo(aa[0]), bfhi(aa[0]), bflo(aa[1]), bfhi(aa[1])};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(add_nc(add_nc(mul_nc(acc[ct][pt][4 * g + e], sc[e]), bi[e]), ad[e]), 0.f);
            const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(cell) = o;
          }
        }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < NSTO; ++i) {
        int t, c;
        unsigned pix;
        sto_item(i, t, c, pix);
        sto_r[i] = *reinterpret_cast<const u32x4*>(S + t * ROW + ((c ^ (t & 15)) << 4));
      }
    }
    store_rows(rs_out, (unsigned)p.ldo, (unsigned)(3 * P));
  }
}

template <int P, int TY, int TX>
__global__ __launch_bounds__(512) void bneck_fwd_kernel(const BnK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bneck_fwd_body<P, TY, TX>(p, smem);
}

}  // namespace

extern "C" int dsl_bottleneck_fwd_supported(const dsl_bneck_desc* d) {
  if (!d) return 0;
  if (d->planes != 128 && d->planes != 256) return 0;
  if (d->cin % 64 || d->cin <= 0 || d->ldx < d->cin || d->ldx % 8 || d->ldi % 8 || d->ldo % 8 || d->ldi < 4 * d->planes || d->ldo < 4 * d->planes) return 0;
  if (d->stride != 1 && d->stride != 2) return 0;
  if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->hin < (d->h - 1) * d->stride + 1 || d->win < (d->w - 1) * d->stride + 1) return 0;
  if ((long long)d->n * d->hin * d->win * d->ldx * 2 >= 0x7fff0000LL || (long long)d->n * d->h * d->w * d->ldi * 2 >= 0x7fff0000LL) return 0;
  // a1 / a2 / out leave through 32-bit buffer offsets in 16-byte chunks as well: an `out` extent past 2 GiB (ldo > ldi)
  // would be dropped silently by the out-of-range rule, a pointer off a 16-byte boundary would fault
  if ((long long)d->n * d->h * d->w * d->ldo * 2 >= 0x7fff0000LL) return 0;
  const void* ptrs[] = {d->x, d->idt, d->a1, d->a2, d->out, d->w1, d->w2, d->w3};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q & 15)) return 0;
  return 1;
}

extern "C" int dsl_bottleneck_fwd(const dsl_bneck_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_bottleneck_fwd: null descriptor");
  DSL_CHECK(dsl_bottleneck_fwd_supported(d), "dsl_bottleneck_fwd: unsupported shape (planes %d cin %d stride %d ldx %d ldi %d ldo %d)", d->planes,
            d->cin, d->stride, d->ldx, d->ldi, d->ldo);
  DSL_CHECK(d->x && d->w1 && d->w2 && d->w3 && d->idt && d->a1 && d->a2 && d->out && d->s1 && d->b1 && d->s2 && d->b2 && d->s3 && d->b3,
            "dsl_bottleneck_fwd: null tensor pointer");
  BnK k;
  memset(&k, 0, sizeof(k));
  k.x = (const uint16_t*)d->x; k.w1 = (const uint16_t*)d->w1; k.w2 = (const uint16_t*)d->w2; k.w3 = (const uint16_t*)d->w3;
  k.idt = (const uint16_t*)d->idt;
  k.s1 = d->s1; k.b1 = d->b1; k.s2 = d->s2; k.b2 = d->b2; k.s3 = d->s3; k.b3 = d->b3;
  k.a1 = (uint16_t*)d->a1; k.a2 = (uint16_t*)d->a2; k.out = (uint16_t*)d->out;
  k.n = d->n; k.hin = d->hin; k.win = d->win; k.h = d->h; k.w = d->w;
  k.cin = d->cin; k.ldx = d->ldx; k.stride = d->stride; k.ldi = d->ldi; k.ldo = d->ldo;
  k.dbg = 0;
#ifdef DSL_ABLATE_BUILD
  { const char* e = getenv("DSL_BNECK_DBG"); k.dbg = e ? atoi(e) : 0; }
#endif
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)d->n * d->h * d->w, P = d->planes;
  int prof = -1;
  if (dsl_prof_active())
    prof = dsl_prof_begin(2, 2.0 * px * (P * d->cin + 9.0 * P * P + 4.0 * P * P), st,
                          2.0 * (px * d->stride * d->stride * d->cin + px * (2 * P + 8 * P) + P * d->cin + 13.0 * P * P));
#define LAUNCHB(P_, TY_, TX_)                                                                                                  \
  do {                                                                                                                         \
    typedef Cfg<P_, TY_, TX_> KC_;                                                                                             \
    k.tiles_y = (d->h + TY_ - 1) / TY_;                                                                                        \
    k.tiles_x = (d->w + TX_ - 1) / TX_;                                                                                        \
    k.ntiles = d->n * k.tiles_y * k.tiles_x;                                                                                   \
    k.xcd_chunk = (k.ntiles + 7) / 8;                                                                                          \
    static bool attr_ = false;                                                                                                 \
    if (!attr_) {                                                                                                              \
      hipFuncSetAttribute((const void*)bneck_fwd_kernel<P_, TY_, TX_>, hipFuncAttributeMaxDynamicSharedMemorySize, KC_::LDS);  \
      attr_ = true;                                                                                                            \
    }                                                                                                                          \
    hipLaunchKernelGGL((bneck_fwd_kernel<P_, TY_, TX_>), dim3(8 * k.xcd_chunk), dim3(512), KC_::LDS, st, k);                   \
  } while (0)
  if (d->planes == 128) LAUNCHB(128, 12, 16);
  else LAUNCHB(256, 5, 12);
#undef LAUNCHB
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("bneck_fwd_kernel");
  return 0;
}
