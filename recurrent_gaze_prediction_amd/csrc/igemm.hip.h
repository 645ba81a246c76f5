// Implicit-GEMM convolution core for gfx950 (MI355X, CDNA4).
//
// One kernel template serves every dense contraction on the gaze path: the
// 1024->P projection, the hoisted ConvGRU input convolutions, the per-step
// recurrent convolutions (gate math fused in the epilogue), the transposed
// convolutions of the saliency head (as gather-form sub-pixel phases) and the
// C3D 3x3x3 convolutions (bias + ReLU + max-pool fused in the epilogue).
//
//   Out[m, n] = sum_k A[m, k] * Wp[n, k]
//
// * A is never materialised.  Activations live in HBM as halo-padded
//   channels-last images, so row m's operand is   A + rowin[m] + koff[kt]
//   with no bounds checks: rowin comes from a per-image offset table (any row
//   order -- pooling-window-major for the fused max-pool, phase-major for the
//   transposed convs), koff from a per-K-chunk table of tap offsets.
// * Wp is the filter pre-packed as [N][K] (K contiguous), so both MFMA operands
//   are 16-byte K-contiguous fragments.
// * Both tiles are staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//   instruction = 8 rows x 128 B) into a double buffer.  LDS images are
//   lane-linear, so the bank-conflict XOR swizzle (16-B chunk ^= row&7) is
//   applied on the per-lane SOURCE address and again on the ds_read_b128.
// * T = bf16 : v_mfma_f32_16x16x32_bf16, 64 elements per 128-B K-chunk.
//   T = f32  : v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), 32 per chunk.
//   Accumulation is always fp32.
// * The epilogue round-trips the accumulators through LDS one wave-row slab at
//   a time so that every thread owns 8 consecutive output channels of one output
//   row (16-B coalesced stores) and can reduce the P rows of a pooling window.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rgp {

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int BKE = 32;  // elements per 128-byte K-chunk
  static constexpr int EPC = 4;   // elements per 16-byte LDS chunk
  static __device__ __forceinline__ float to(float v) { return v; }
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int BKE = 64;
  static constexpr int EPC = 8;
  static __device__ __forceinline__ bf16_t to(float v) { return f2bf(v); }
  static __device__ __forceinline__ float from(bf16_t v) { return bf2f(v); }
};

struct IgemmParams {
  const void* A;             // activation base (T)
  const void* W;             // packed filter [Npad][K] (T), Npad multiple of 128
  const int* in_tab;         // [Mw] element offset of row's receptive-field origin inside one image
  const int* koff;           // [nk*G] element offset added for each K-(sub)chunk
  long long in_img_stride;   // elements between images
  int Mw;                    // rows per image
  int M;                     // total rows
  int N;                     // real output channels
  int K;                     // packed K (multiple of BKE)
  int nk;                    // K / BKE
  int ntile_group = 0;       // staggered kernel: row tiles per column-tile run (0 / 1 = column tiles innermost)
  int pool_regs = 1;         // staggered kernel, P = 8 without arg-max: pool in registers before staging (RGP_POOLREGS)
  int tile128 = 0;           // host side only (launch_igemm): 1 = never the 256-row persistent kernels (RGP_C3D_KERNELS_TILE128)
};

// Epilogue operands (superset; each functor reads what it needs).
struct EpiParams {
  void* out;                 // primary output
  const int* out_tab;        // [Mw/P] element offset of an output row inside one output image
  long long out_img_stride;  // elements
  long long out_extra;       // extra element offset (box origin in conv3d_halo_kernel)
  const float* bias;         // [N]
  // ConvGRU gate epilogues (gaze_grcn.py:118-127)
  const float* xpre;         // hoisted W*x pre-activations for this step, [img][49][xpre_ld]
  long long xpre_img_stride;
  int xpre_ld;
  int xpre_col;              // column of this kernel's first gate inside xpre
  int S;                     // state channels
  int state_rows;            // rows per state image (49)
  const float* h_prev;       // [img][Mw][S] fp32 state h_{t-1}
  float* h_next;             // [img][Mw][S] fp32 state h_t
  float* u_gate;             // [img][Mw][S] update gate (written by ZR, read by C)
  float* r_save;             // optional [img][Mw][S]
  float* c_save;             // optional [img][Mw][S]
  void* out2;                // C epilogue: batch-normalised h_t for the head
  const int* out2_tab;
  long long out2_img_stride;
  long long out2_img_mul;    // head image index = img*out2_img_mul + out2_img_add  (b*T + t)
  long long out2_img_add;
  const float* bn_gamma;     // [S] of this timestep
  const float* bn_beta;
  float bn_inv_std;
  // training forward of pooled conv layers: index (0..P-1, window order) of the first maximum of every
  // pooling window, [img][Mw/P][N] bytes; null = not recorded
  unsigned char* argmax;
  const void* mask;          // EpiStoreMask: forward output whose sign gates the gradient
  // EpiReluMaxout: training-time dropout between the ReLU and the maxout (gaze_grcn_cascade.py:401-402):
  // keep bytes [img][N] in the layer's natural unit order (null = off), kept values * drop_inv_keep
  const unsigned char* drop_mask;
  float drop_inv_keep;
};

// ---------------------------------------------------------------------------
// Epilogue functors.  apply() gets 8 consecutive output channels n0..n0+7 of one
// (pooled) output row; img/ml identify the row (ml = row index inside the image,
// already divided by the pooling factor).
// ---------------------------------------------------------------------------
template <typename TO> __device__ __forceinline__ void store8(TO* dst, const float* v, int nvalid);
template <> __device__ __forceinline__ void store8<float>(float* dst, const float* v, int nvalid) {
  if (nvalid >= 8) {
    *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else {
    for (int i = 0; i < nvalid; ++i) dst[i] = v[i];
  }
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* dst, const float* v, int nvalid) {
  if (nvalid >= 8) {
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
    *(u32x4*)dst = pk;
  } else {
    for (int i = 0; i < nvalid; ++i) dst[i] = f2bf(v[i]);
  }
}

// Element offset of output row (img, ml) inside e.out.  The staggered kernel resolves it once per row in its
// prologue (a table lookup per item in the epilogue is a dependent global load per slab) and hands it to apply_at().
__device__ __forceinline__ long long epi_out_base(const EpiParams& e, int img, int ml) {
  return (long long)img * e.out_img_stride + e.out_extra + e.out_tab[ml];
}

// out = [relu](acc + bias) stored as TO.
template <typename TO, bool BIAS, bool RELU> struct EpiStore {
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    apply_at(e, N, img, ml, epi_out_base(e, img, ml), n0, v);
  }
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int, int, long long base, int n0, float* v) {
    const int nvalid = N - n0;
    if (nvalid <= 0) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (BIAS) v[i] += (i < nvalid) ? e.bias[n0 + i] : 0.f;
      if (RELU) v[i] = fmaxf(v[i], 0.f);
    }
    store8<TO>((TO*)e.out + base + n0, v, nvalid);
  }
};

// A kernel that has the bias of its 8 columns in registers adds it itself and calls NoBias::apply_at.
template <class E> struct EpiBiasSplit { static constexpr bool value = false; using NoBias = E; };
template <typename TO, bool RELU> struct EpiBiasSplit<EpiStore<TO, true, RELU>> {
  static constexpr bool value = true;
  using NoBias = EpiStore<TO, false, RELU>;
};

// dgrad through a ReLU: out = acc where the layer's forward output (e.mask, same geometry as out)
// was positive, else 0.
template <typename TO> struct EpiStoreMask {
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    apply_at(e, N, img, ml, epi_out_base(e, img, ml), n0, v);
  }
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int, int, long long base, int n0, float* v) {
    const int nvalid = N - n0;
    if (nvalid <= 0) return;
    const long long idx = base + n0;
    const TO* m = (const TO*)e.mask + idx;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nvalid && !(Elem<TO>::from(m[i]) > 0.f)) v[i] = 0.f;
    store8<TO>((TO*)e.out + idx, v, nvalid);
  }
};

// ShallowNet fully-connected layers (saliency_shallownet.py:139-185): relu(acc + bias) then
// maxout over the two halves of the layer.  The filter is packed with the halves
// interleaved (packed column 2j = unit j, 2j+1 = unit j + N/2), so the maxout partner of a
// column is its neighbour and 8 packed columns give 4 outputs at column n0/2.
template <typename TO> struct EpiReluMaxout {
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    apply_at(e, N, img, ml, epi_out_base(e, img, ml), n0, v);
  }
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int img, int, long long base, int n0, float* v) {
    const int nvalid = N - n0;
    if (nvalid <= 0) return;
    TO* dst = (TO*)e.out + base + (n0 >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (2 * i + 1 < nvalid) {
        float a = fmaxf(v[2 * i] + e.bias[n0 + 2 * i], 0.f), b = fmaxf(v[2 * i + 1] + e.bias[n0 + 2 * i + 1], 0.f);
        if (e.drop_mask) {    // packed column 2j = unit j, 2j+1 = unit j + N/2
          const unsigned char* dm = e.drop_mask + (long long)img * N + (n0 >> 1) + i;
          a = dm[0] ? a * e.drop_inv_keep : 0.f;
          b = dm[N >> 1] ? b * e.drop_inv_keep : 0.f;
        }
        dst[i] = Elem<TO>::to(fmaxf(a, b));
        // training: which half carries the gradient (tf.maximum sends ties to the first operand), 0 = ReLU-gated
        if (e.argmax) e.argmax[(long long)img * (N >> 1) + (n0 >> 1) + i] = fmaxf(a, b) > 0.f ? (a >= b ? 1 : 2) : 0;
      }
    }
  }
};

// out (fp32) += acc : the second contribution to the carried state gradient in BPTT.
struct EpiAccumF32 {
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    apply_at(e, N, img, ml, epi_out_base(e, img, ml), n0, v);
  }
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int, int, long long base, int n0, float* v) {
    const int nvalid = N - n0;
    if (nvalid <= 0) return;
    float* dst = (float*)e.out + base + n0;
    for (int i = 0; i < 8 && i < nvalid; ++i) dst[i] += v[i];
  }
};

// out (fp32) += acc with global float atomics: split-K partial sums of the wgrad GEMMs.
struct EpiAtomicAddF32 {
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    apply_at(e, N, img, ml, epi_out_base(e, img, ml), n0, v);
  }
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int, int, long long base, int n0, float* v) {
    const int nvalid = N - n0;
    if (nvalid <= 0) return;
    float* dst = (float*)e.out + base + n0;
    for (int i = 0; i < 8 && i < nvalid; ++i) atomicAdd(dst + i, v[i]);
  }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float e = expf(-2.f * fabsf(x));
  const float t = (1.f - e) / (1.f + e);
  return copysignf(t, x);
}

// Update / reset gates:  u = sigmoid(Wz*x + Uz*h), r = sigmoid(Wr*x + Ur*h)
// (gaze_grcn.py:108-119).  Columns [0,S) are z, [S,2S) are r.  Writes u (fp32)
// and the candidate conv's operand r*h (T, halo-padded state image).
template <typename T> struct EpiGruZR {
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int img, int ml, long long, int n0, float* v) {
    apply(e, N, img, ml, n0, v);
  }
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    if (n0 >= N) return;
    const float* xp = e.xpre + (long long)img * e.xpre_img_stride + (long long)ml * e.xpre_ld + e.xpre_col + n0;
    const f32x4 x0 = *(const f32x4*)xp, x1 = *(const f32x4*)(xp + 4);
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = sigmoidf_(v[i] + (i < 4 ? x0[i] : x1[i - 4]));
    const long long srow = ((long long)img * e.state_rows + ml) * e.S;
    if (n0 < e.S) {
      store8<float>(e.u_gate + srow + n0, g, 8);
    } else {
      const int c0 = n0 - e.S;
      const f32x4 h0 = *(const f32x4*)(e.h_prev + srow + c0), h1 = *(const f32x4*)(e.h_prev + srow + c0 + 4);
      if (e.r_save) store8<float>(e.r_save + srow + c0, g, 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] *= (i < 4 ? h0[i] : h1[i - 4]);
      store8<T>((T*)e.out + (long long)img * e.out_img_stride + e.out_tab[ml] + c0, g, 8);
    }
  }
};

// Candidate + blend:  c = tanh(W*x + U*(r.h)),  h' = u*h + (1-u)*c
// (gaze_grcn.py:122-127), then the per-timestep inference batch-norm
// (gaze_grcn.py:325) for the head.  Writes h' (fp32 state), h' (T, padded, next
// step's operand) and BN(h') (T, padded head image of frame img*mul+add = b*T+t).
template <typename T> struct EpiGruC {
  static __device__ __forceinline__ void apply_at(const EpiParams& e, int N, int img, int ml, long long, int n0, float* v) {
    apply(e, N, img, ml, n0, v);
  }
  static __device__ __forceinline__ void apply(const EpiParams& e, int N, int img, int ml, int n0, float* v) {
    if (n0 >= N) return;
    const float* xp = e.xpre + (long long)img * e.xpre_img_stride + (long long)ml * e.xpre_ld + e.xpre_col + n0;
    const f32x4 x0 = *(const f32x4*)xp, x1 = *(const f32x4*)(xp + 4);
    const long long srow = ((long long)img * e.state_rows + ml) * e.S + n0;
    const f32x4 u0 = *(const f32x4*)(e.u_gate + srow), u1 = *(const f32x4*)(e.u_gate + srow + 4);
    const f32x4 h0 = *(const f32x4*)(e.h_prev + srow), h1 = *(const f32x4*)(e.h_prev + srow + 4);
    float c[8], hn[8], hb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float u = i < 4 ? u0[i] : u1[i - 4];
      const float h = i < 4 ? h0[i] : h1[i - 4];
      c[i] = tanhf_(v[i] + (i < 4 ? x0[i] : x1[i - 4]));
      hn[i] = u * h + (1.f - u) * c[i];
      hb[i] = e.bn_gamma[n0 + i] * (hn[i] * e.bn_inv_std) + e.bn_beta[n0 + i];
    }
    if (e.c_save) store8<float>(e.c_save + srow, c, 8);
    store8<float>(e.h_next + srow, hn, 8);
    store8<T>((T*)e.out + (long long)img * e.out_img_stride + e.out_tab[ml] + n0, hn, 8);
    const long long himg = (long long)img * e.out2_img_mul + e.out2_img_add;
    store8<T>((T*)e.out2 + himg * e.out2_img_stride + e.out2_tab[ml] + n0, hb, 8);
  }
};

// ---------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void step(f32x4& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void step(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
  }
};

// max over the P rows of one pooling window (8 adjacent columns of the fp32 slab in LDS)
template <int P> __device__ __forceinline__ void pool_window(const float* src, int ld, float* v) {
  f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
#pragma unroll
  for (int q = 1; q < P; ++q) {
    const f32x4 w0 = *(const f32x4*)(src + q * ld), w1 = *(const f32x4*)(src + q * ld + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], w0[i]); v1[i] = fmaxf(v1[i], w1[i]); }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = v0[i]; v[4 + i] = v1[i]; }
}
// same, also recording which row held the (first) maximum: 8 bytes at amax
template <int P> __device__ __forceinline__ void pool_window_argmax(const float* src, int ld, float* v, unsigned char* amax) {
  unsigned idx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = src[i]; idx[i] = 0; }
#pragma unroll
  for (int q = 1; q < P; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float w = src[q * ld + i];
      if (w > v[i]) { v[i] = w; idx[i] = q; }
    }
  const unsigned lo = idx[0] | (idx[1] << 8) | (idx[2] << 16) | (idx[3] << 24);
  const unsigned hi = idx[4] | (idx[5] << 8) | (idx[6] << 16) | (idx[7] << 24);
  *(uint2*)amax = make_uint2(lo, hi);
}

template <int BM, int BN> struct IgemmSmem {
  static constexpr int TILE_BYTES = (BM + BN) * 128;
  static constexpr int ROWINFO_OFF = 2 * TILE_BYTES;
  static constexpr int BYTES = ROWINFO_OFF + BM * 24;   // int64 rowin + int img + int ml + int64 rowout per row
};

// G = K-subchunks per 128-byte chunk that carry their own tap offset (1 for
// Cin*sizeof(T) >= 128 B; 2/4 when a 128-B chunk spans several taps).
// P = rows per pooling window (1, 4 or 8), rows of a window are consecutive in m.
template <typename T, int BM, int BN, int WM, int WN, int G, int P, class Epi>
__device__ __forceinline__ void igemm_tile(const IgemmParams& p, const EpiParams& e) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int A_PER_WAVE = (BM / 8) / NW, B_PER_WAVE = (BN / 8 + NW - 1) / NW;
  constexpr int ESZ = sizeof(T);
  constexpr int TILE_BYTES = IgemmSmem<BM, BN>::TILE_BYTES;
  static_assert((BM / 8) % NW == 0, "A groups must divide over waves");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
  static_assert(WTM % P == 0, "pool window inside a slab");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* s_rowin = (long long*)(smem + IgemmSmem<BM, BN>::ROWINFO_OFF);
  int* s_rowimg = (int*)(s_rowin + BM);
  int* s_rowml = s_rowimg + BM;
  long long* s_rowout = (long long*)(s_rowml + BM);   // output offset of the row's pooling window, resolved here
                                                      // so that no epilogue item waits on a table look-up

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each
  // XCD a contiguous run of tiles so halo rows / weight panels are L2 hits.
  const int n_nt = (p.N + BN - 1) / BN;
  const int n_mt = (p.M + BM - 1) / BM;
  const int nwg = n_mt * n_nt;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, y = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int mt = bid / n_nt, nt = bid % n_nt;
  const int m0 = mt * BM, n0 = nt * BN;

  // K-chunk offsets are fetched one iteration ahead of their use (a dependent load at
  // the top of the iteration would expose a full L2 round trip before the DMA issues).
  const int asub_ = (((tid & 7) ^ ((tid & 63) >> 3))) / (8 / G);
  auto load_koff = [&](int kt) -> int {
    kt = kt < p.nk ? kt : p.nk - 1;
    return G == 1 ? p.koff[kt] : p.koff[kt * G + asub_];
  };
  // split-K (gridDim.y > 1, wgrad GEMMs): this block reduces K-chunks [kt_begin, kt_end)
  const int kt_begin = (int)((long long)p.nk * blockIdx.y / gridDim.y);
  const int kt_end = (int)((long long)p.nk * (blockIdx.y + 1) / gridDim.y);
  const int ko_first = load_koff(kt_begin);   // issued before the row-table chain so the latencies overlap
  int ko_next = load_koff(kt_begin + 1);

  for (int r = tid; r < BM; r += NT) {
    int m = m0 + r;
    const bool valid = m < p.M;
    if (!valid) m = p.M - 1;
    const int img = m / p.Mw, ml = m - img * p.Mw;
    s_rowin[r] = (long long)img * p.in_img_stride + p.in_tab[ml];
    s_rowimg[r] = valid ? img : -1;
    s_rowml[r] = ml;
    s_rowout[r] = (valid && r % P == 0) ? epi_out_base(e, img, ml / P) : 0;
  }
  __syncthreads();

  // Per-lane source bases.  A wave instruction fills 8 rows x 128 B: lane l ->
  // row l>>3, physical 16-B slot l&7, which holds logical chunk (l&7)^(l>>3).
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const char* a_src[A_PER_WAVE];
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int r = (wave * A_PER_WAVE + j) * 8 + lrow;
    a_src[j] = (const char*)p.A + s_rowin[r] * ESZ + (lchunk % (8 / G)) * 16;
  }
  const char* b_src[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int r = (wave * B_PER_WAVE + j) * 8 + lrow;
    b_src[j] = (const char*)p.W + ((long long)(n0 + r) * p.K) * ESZ + lchunk * 16;
  }
  const bool b_active = (BN / 8 >= NW) || (wave < BN / 8);

  auto stage = [&](int buf, int kt, int koff_elems) {
    char* abuf = smem + buf * TILE_BYTES;
    char* bbuf = abuf + BM * 128;
    const long long ko = (long long)koff_elems * ESZ;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + ko),
                                       (__attribute__((address_space(3))) void*)(abuf + (wave * A_PER_WAVE + j) * 1024),
                                       16, 0, 0);
    }
    if (b_active) {
      const long long kb = (long long)kt * 128;
#pragma unroll
      for (int j = 0; j < B_PER_WAVE; ++j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                         (__attribute__((address_space(3))) void*)(bbuf + (wave * B_PER_WAVE + j) * 1024),
                                         16, 0, 0);
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int buf) {
    const char* abuf = smem + buf * TILE_BYTES + (wm * WTM + frow) * 128;
    const char* bbuf = smem + buf * TILE_BYTES + BM * 128 + (wn * WTN + frow) * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int pc = ((s * 4 + fk) ^ (frow & 7)) * 16;
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(abuf + i * 16 * 128 + pc);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *(const f32x4*)(bbuf + j * 16 * 128 + pc);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mma<T>::step(acc[i][j], a[i], b[j]);
    }
  };

  stage(0, kt_begin, ko_first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  // One loop, no peeled tail: with a separate tail the compiler rotates the accumulator
  // registers through ~70 v_accvgpr moves per iteration.
#pragma clang loop unroll(disable)
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) {
      stage(cur ^ 1, kt + 1, ko_next);
      ko_next = load_koff(kt + 2);
    }
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: one wave-row slab (WTM rows x BN cols, fp32) at a time ----
  constexpr int LDS_LD = BN + 4;
  float* stg = (float*)smem;
  static_assert(WTM * LDS_LD * 4 <= 2 * TILE_BYTES, "staging fits in the tile buffers");
  constexpr int CG = BN / 8;
  constexpr int ITEMS = (WTM / P) * CG;
  // a thread keeps its column group over all items when NT % CG == 0: its bias is fetched once, up front
  using Bias = EpiBiasSplit<Epi>;
  constexpr bool HOIST = Bias::value && (NT % CG == 0);
  float bias8[8];
  if constexpr (HOIST) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bias8[i] = (n0 + (tid % CG) * 8 + i < p.N) ? e.bias[n0 + (tid % CG) * 8 + i] : 0.f;
  }
#pragma unroll 1
  for (int slab = 0; slab < WM; ++slab) {
    if (wm == slab) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stg[(i * 16 + fk * 4 + r) * LDS_LD + wn * WTN + j * 16 + frow] = acc[i][j][r];
    }
    __syncthreads();
    for (int it = tid; it < ITEMS; it += NT) {
      const int g = it / CG, cg = it - g * CG;
      const int rt = slab * WTM + g * P;
      const int img = s_rowimg[rt];
      if (img >= 0) {
        float v[8];
        const float* src = stg + (g * P) * LDS_LD + cg * 8;
        if (P > 1 && e.argmax && n0 + cg * 8 < p.N) pool_window_argmax<P>(src, LDS_LD, v, e.argmax + ((long long)img * (p.Mw / P) + s_rowml[rt] / P) * p.N + n0 + cg * 8);
        else pool_window<P>(src, LDS_LD, v);
        if constexpr (HOIST) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += bias8[i];
          Bias::NoBias::apply_at(e, p.N, img, s_rowml[rt] / P, s_rowout[rt], n0 + cg * 8, v);
        } else {
          Epi::apply_at(e, p.N, img, s_rowml[rt] / P, s_rowout[rt], n0 + cg * 8, v);
        }
      }
    }
    __syncthreads();
  }
}

template <typename T, int BM, int BN, int WM, int WN, int G, int P, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmParams p, const EpiParams e) {
  igemm_tile<T, BM, BN, WM, WN, G, P, Epi>(p, e);
}

// Several problems of one tile shape in one launch (blockIdx.z = problem; the transposed convolution's sub-pixel phases:
// 49 GEMMs of 215 tiles each, which one by one leave CUs idle and pay 49 launch latencies).  Parameters come from device
// arrays; blocks beyond a problem's own tile count leave at once.
template <typename T, int BM, int BN, int WM, int WN, int G, int P, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void igemm_grouped_kernel(const IgemmParams* __restrict__ ps, const EpiParams* __restrict__ es,
                                                                    long long a_bytes, long long out_bytes) {
  IgemmParams p = ps[blockIdx.z];
  if ((int)blockIdx.x >= ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN)) return;
  EpiParams e = es[blockIdx.z];
  p.A = (const char*)p.A + a_bytes;              // the same problems on another slice of the operands (one time step)
  e.out = (char*)e.out + out_bytes;
  igemm_tile<T, BM, BN, WM, WN, G, P, Epi>(p, e);
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny GEMM: ONE row tile (M <= 64) and a wide output -- the fc-GRU recurrence (M = B = 64 rows, N = 3248 / 1624, K = 1624,
// /root/reference/models/gaze_rnn.py:315-349).  A block owns 16 output columns of all 64 rows, so the columns spread over
// N / 16 = 203 / 102 CUs; the K reduction is split over the 16 waves of the block -- wave w = (row tile w & 3, K slice
// w >> 2) -- so that a barrier interval covers FOUR K-tiles (13 intervals for K = 1624 instead of 51) and four waves per
// SIMD hide each other's LDS latency.  Three LDS stages of 4 x (64 + 16) x 128 B, two in flight behind a counted vmcnt; the
// K-offset table comes through the scalar cache.  The four slices are summed through LDS before the epilogue functor runs
// (same functors as igemm_kernel).  G = 1, P = 1.
struct SkinnySmem {
  static constexpr int KT = 4;                              // K-tiles per stage
  static constexpr int TILE = (64 + 16) * 128;              // one K-tile: 64 A rows + 16 B rows of 128 B
  static constexpr int STAGE = KT * TILE;                   // 40 960
  static constexpr int NS = 3;
  static constexpr int RED_OFF = NS * STAGE;                // [4 slices][64 rows][20] fp32
  static constexpr int ROWINFO_OFF = RED_OFF + 4 * 64 * 20 * 4;
  static constexpr int BYTES = ROWINFO_OFF + 64 * 24;       // 144 896
};

template <typename T, class Epi>
__global__ __launch_bounds__(1024) void igemm_skinny_kernel(const IgemmParams p, const EpiParams e) {
  using S = SkinnySmem;
  constexpr int ESZ = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* s_rowin = (long long*)(smem + S::ROWINFO_OFF);
  int* s_rowimg = (int*)(s_rowin + 64);
  int* s_rowml = s_rowimg + 64;
  long long* s_rowout = (long long*)(s_rowml + 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wmt = wave & 3, wq = wave >> 2;                 // row tile, K slice
  const int n0 = blockIdx.x * 16;
  auto load_koff = [&](int kt) -> int {
    kt = kt < p.nk ? kt : p.nk - 1;
    return ((const __attribute__((address_space(4))) int*)p.koff)[kt];
  };
  for (int r = tid; r < 64; r += 1024) {
    int m = r;
    const bool valid = m < p.M;
    if (!valid) m = p.M - 1;
    const int img = m / p.Mw, ml = m - img * p.Mw;
    s_rowin[r] = (long long)img * p.in_img_stride + p.in_tab[ml];
    s_rowimg[r] = valid ? img : -1;
    s_rowml[r] = ml;
    s_rowout[r] = valid ? epi_out_base(e, img, ml) : 0;
  }
  __syncthreads();
  // DMA sources: a wave instruction fills 8 rows x 128 B (lane -> row lane >> 3, physical chunk lane & 7 = logical chunk
  // (lane & 7) ^ (lane >> 3)); per stage 4 K-tiles x (8 A groups + 2 B groups): wave w stages A groups 2w, 2w+1 of the 32
  // (K-tile = group >> 3), waves 0..7 also B group w of the 8
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const char* a_src[2];
  int a_kt[2], a_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = 2 * wave + j;
    a_kt[j] = g >> 3;
    a_src[j] = (const char*)p.A + s_rowin[(g & 7) * 8 + lrow] * ESZ + lchunk * 16;
    a_dst[j] = a_kt[j] * S::TILE + (g & 7) * 1024;
  }
  const bool b_active = wave < 8;
  const int b_kt = wave >> 1;
  const char* b_src = (const char*)p.W + ((long long)(n0 + (wave & 1) * 8 + lrow) * p.K) * ESZ + lchunk * 16;
  const int b_dst = b_kt * S::TILE + 64 * 128 + (wave & 1) * 1024;
  const int n_it = (p.nk + S::KT - 1) / S::KT;              // barrier intervals; K-tiles beyond nk are clamped duplicates, never multiplied
  auto stage = [&](int slot, int it, int ko0, int ko1) {
    char* base = smem + slot * S::STAGE;
    {
      const int kt = it * S::KT + a_kt[0];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[0] + (long long)ko0 * ESZ),
                                       (__attribute__((address_space(3))) void*)(base + a_dst[0]), 16, 0, 0);
      (void)kt;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[1] + (long long)ko1 * ESZ),
                                     (__attribute__((address_space(3))) void*)(base + a_dst[1]), 16, 0, 0);
    if (b_active) {
      int kt = it * S::KT + b_kt;
      if (kt >= p.nk) kt = p.nk - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + (long long)kt * 128),
                                       (__attribute__((address_space(3))) void*)(base + b_dst), 16, 0, 0);
    }
  };
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int slot, int it) {
    if (it * S::KT + wq >= p.nk) return;                    // (wave-uniform) K-tile past the end
    const char* abuf = smem + slot * S::STAGE + wq * S::TILE + (wmt * 16 + frow) * 128;
    const char* bbuf = smem + slot * S::STAGE + wq * S::TILE + 64 * 128 + frow * 128;
    const int pc0 = ((fk) ^ (frow & 7)) * 16, pc1 = ((4 + fk) ^ (frow & 7)) * 16;
    const f32x4 a0 = *(const f32x4*)(abuf + pc0), b0 = *(const f32x4*)(bbuf + pc0);
    const f32x4 a1 = *(const f32x4*)(abuf + pc1), b1 = *(const f32x4*)(bbuf + pc1);
    Mma<T>::step(acc0, a0, b0);
    Mma<T>::step(acc1, a1, b1);
  };
  // K offsets of this wave's two A groups for interval `it`
  auto ko_of = [&](int it, int j) { return load_koff(it * S::KT + a_kt[j]); };
  // prologue: intervals 0 and 1
  stage(0, 0, ko_of(0, 0), ko_of(0, 1));
  if (n_it > 1) stage(1, 1, ko_of(1, 0), ko_of(1, 1));
  int k0 = ko_of(2, 0), k1 = ko_of(2, 1);
  int slot = 0, fill = 2;
#pragma clang loop unroll(disable)
  for (int it = 0; it < n_it; ++it) {
    // interval `it` landed: younger is (at most) interval it + 1
    if (it + 1 < n_it) {
      if (b_active) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (it + 2 < n_it) {
      stage(fill, it + 2, k0, k1);
      k0 = ko_of(it + 3, 0); k1 = ko_of(it + 3, 1);
    }
    compute(slot, it);
    slot = slot + 1 == S::NS ? 0 : slot + 1;
    fill = fill + 1 == S::NS ? 0 : fill + 1;
  }
  // ---- reduce the four K slices, then the epilogue functor on (row, 8 columns) items ----
  acc0 += acc1;
  float* red = (float*)(smem + S::RED_OFF);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wq * 64 + wmt * 16 + fk * 4 + r) * 20 + frow] = acc0[r];
  __syncthreads();
  if (tid < 128) {
    const int row = tid >> 1, cgp = tid & 1;
    const int img = s_rowimg[row];
    if (img >= 0) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o = row * 20 + cgp * 8 + i;
        v[i] = red[o] + red[64 * 20 + o] + red[2 * 64 * 20 + o] + red[3 * 64 * 20 + o];
      }
      Epi::apply_at(e, p.N, img, s_rowml[row], s_rowout[row], n0 + cgp * 8, v);
    }
  }
}

}  // namespace rgp
