// C3D conv1a (3x3x3, 3->64, pad 1) + bias + ReLU + pool1 (1x2x2 max) for gfx950, bf16.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-66.
//
// conv1a has K = 81: far too short for the LDS-staged implicit-GEMM tile loop (three K-chunks per tile, so
// tile set-up, barriers and the epilogue dominate), and its output (6.4 MB per window) is 5x its input, so the
// kernel is organised around streaming:
//
//  * input  act0 [n][18][114][116][4] bf16 (halo-padded, channels 3->4, 8 bytes per pixel).
//  * a block (4 waves) walks JOBS = (window, plane z, 4 pooled rows): the 3 x 10 x 116-pixel input patch of a job
//    (27.2 KiB) is fetched ONCE, with coalesced 16-byte loads into registers while the previous job computes,
//    and parked in LDS (double-buffered, one barrier per job).  The first version of this kernel built its
//    A-fragments straight from global memory: every input pixel went through the texture path ~50 times
//    (64 GB per 1024 windows at 64 B/clk/CU = as long as the MFMAs themselves) and the kernel ran at 31 %
//    MFMA utilisation; fragments now come from LDS (two ds_read_b64 each).
//  * K is ordered (kz,ky,kx | c4): 27 taps x 4 channels = 108, padded to 128 = 4 MFMA k-steps (was 160 = 5).
//    A lane's 8-element A-fragment is two taps = two 8-byte pixels.
//  * filter packed [64][128] bf16 -> 16 B-fragments (64 VGPRs) loaded once per wave.
//  * wave tile = 32 conv rows = 8 pooled pixels x 64 channels, 28 tiles per job, 7 per wave.  Rows are ordered
//    (pooling window, dy, dx), so the 4 rows of a window are the 4 accumulator registers of one lane
//    (v_mfma_f32_16x16x32 C layout: row = 4*(lane>>4)+reg): pool1 is an in-lane max, then bias + ReLU.
//  * the pooled 8x64 bf16 tile is transposed through 1 KiB of wave-private LDS so the store is one fully
//    coalesced 1 KiB run (8 adjacent pixels x 128 B).
//  * jobs are dealt to the 8 XCDs in contiguous ranges (consecutive workgroup ids sit on different XCDs), so
//    the 3x plane overlap and the y halo between neighbouring jobs are served by one XCD's L2.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct Conv1aParams {
  const float* video;   // FUSED variant: the caller's mean-subtracted clip windows [n][16][112][112][3] fp32, read directly
                        // (the separate video_prep pass -- 2.4 MB read + 1.9 MB written per window, 0.93 ms per 1024
                        // windows at HBM rate -- is folded into the patch fetch)
  const bf16_t* in;     // [n][18][114][116][4]
  const bf16_t* wp;     // [64][128]
  const float* bias;    // [64]
  bf16_t* out;          // [n][18][58][58][64] (halo-padded input of conv2a)
  unsigned char* argmax;  // optional [n][16][56][56][64]: dy*2+dx of the first maximum of each pool1 window
  int n_windows;
};

constexpr int C1_D = 16, C1_H = 112, C1_HP = 114, C1_WP = 116, C1_K = 128;
constexpr int C1_PO = 56;                       // pooled extent
constexpr int C1_XG = C1_PO / 8;                // 8 pooled pixels per wave tile
constexpr int C1_OUT_P = 58;
constexpr int C1_JROWS = 4;                     // pooled rows per job
constexpr int C1_YQ = C1_PO / C1_JROWS;         // 14 jobs per plane
constexpr int C1_JOBS_PER_WINDOW = C1_D * C1_YQ;
constexpr int C1_PROWS = 2 * C1_JROWS + 2;      // 10 input rows per plane of a patch
constexpr int C1_ROWB = C1_WP * 8;              // 928 bytes per input row
constexpr int C1_PATCH = 3 * C1_PROWS * C1_ROWB;          // 27840 bytes
constexpr int C1_CHUNKS = C1_PATCH / 16;                  // 1740 16-byte chunks
constexpr int C1_CPR = C1_ROWB / 16;                      // 58 chunks per row
constexpr int C1_NLD = (C1_CHUNKS + 255) / 256;           // 7 loads per thread
constexpr int C1_TILES = C1_JROWS * C1_XG;                // 28 wave tiles per job
constexpr int C1_SMEM = 2 * C1_PATCH + 4 * 8 * 72 * 2 + 4 * 8 * 72;
// FUSED variant: LDS patch rows of 118 pixel slots (944 B): slot 1 = x -1 (zero), slots 2..113 = x 0..111, so that the
// 4-pixel groups converted from fp32 land on 16-byte boundaries; slots 0, 1 and 114..117 stay zero for the whole kernel
constexpr int C1F_WPL = 118;
constexpr int C1F_PATCH = 3 * C1_PROWS * C1F_WPL * 8;     // 28 320 bytes
constexpr int C1F_GROUPS = 3 * C1_PROWS * 28;             // 840 groups of 4 pixels (12 floats = three 16-byte loads)
constexpr int C1F_NLD = (C1F_GROUPS + 255) / 256;         // 4 per thread
constexpr int C1F_SMEM = 2 * C1F_PATCH + 4 * 8 * 72 * 2 + 4 * 8 * 72;

template <bool FUSED>
static __global__ __launch_bounds__(256, 2) void conv1a_pool_bf16_kernel(const Conv1aParams p) {
  constexpr int WPL = FUSED ? C1F_WPL : C1_WP;             // LDS patch row pitch in pixel slots
  constexpr int XS = FUSED ? 1 : 0;                        // slot of x = -1
  constexpr int PATCH = FUSED ? C1F_PATCH : C1_PATCH;
  extern __shared__ __attribute__((aligned(16))) char c1_smem[];
  char* patch = c1_smem;                                                   // [2][PATCH]
  bf16_t* s_out = (bf16_t*)(c1_smem + 2 * PATCH);                          // per wave: 8 px x (64 ch + 8 pad)
  unsigned char* s_arg = (unsigned char*)(s_out + 4 * 8 * 72);
  if constexpr (FUSED) {      // the x-halo slots are never written afterwards
    for (int i = threadIdx.x; i < 2 * PATCH / 16; i += 256) ((u32x4*)patch)[i] = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, kg = lane >> 4;

  // jobs of this block: XCD x owns the contiguous range [x*per_xcd, (x+1)*per_xcd)
  const long long total = (long long)p.n_windows * C1_JOBS_PER_WINDOW;
  const long long per_xcd = (total + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  long long job = xcd * per_xcd + slot;
  long long job_end = (xcd + 1) * per_xcd;
  if (job_end > total) job_end = total;
  if (job >= job_end) return;

  // filter fragments: B[k = 32s + 8kg + j][n = 16jn + frow]
  f32x4 bfrag[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
      bfrag[s][jn] = *(const f32x4*)(p.wp + (jn * 16 + frow) * C1_K + s * 32 + kg * 8);
  float bias_v[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) bias_v[jn] = p.bias[jn * 16 + frow];

  // patch chunk q = tid + 256 u: where it comes from (elements, relative to the job's first row) and valid?
  int g_off[C1_NLD];
#pragma unroll
  for (int u = 0; u < C1_NLD; ++u) {
    int q = tid + 256 * u;
    if (q >= C1_CHUNKS) q = C1_CHUNKS - 1;                 // duplicate, never written
    const int row = q / C1_CPR, c = q - row * C1_CPR;
    const int kz = row / C1_PROWS, ry = row - kz * C1_PROWS;
    g_off[u] = ((kz * C1_HP + ry) * C1_WP) * 4 + c * 8;
  }
  auto job_src = [&](long long j) -> const bf16_t* {
    const int yq = (int)(j % C1_YQ);
    const int z = (int)((j / C1_YQ) % C1_D);
    const long long n = j / C1_JOBS_PER_WINDOW;
    return p.in + (((n * (C1_D + 2) + z) * C1_HP + yq * 2 * C1_JROWS) * (long long)C1_WP) * 4;
  };
  auto fetch = [&](long long j, u32x4 (&pf)[C1_NLD]) {
    const bf16_t* src = job_src(j);
#pragma unroll
    for (int u = 0; u < C1_NLD; ++u) pf[u] = *(const u32x4*)(src + g_off[u]);
  };
  auto park = [&](char* buf, const u32x4 (&pf)[C1_NLD]) {
#pragma unroll
    for (int u = 0; u < C1_NLD; ++u)
      if (tid + 256 * u < C1_CHUNKS) *(u32x4*)(buf + (tid + 256 * u) * 16) = pf[u];
  };
  // FUSED: group q = tid + 256 u of the patch = row q / 28 (plane kz = row / 10, row ry = row % 10), pixels 4 (q % 28)..+3
  // -> three 16-byte loads of 12 floats from the fp32 window; rows / planes outside the window give zeros
  // (row / group are recomputed from tid where needed: 8 fewer live VGPRs in a kernel at the 256-register limit)
  auto grp = [&](int u, int& row, int& xg) {
    int q = tid + 256 * u;
    if (q >= C1F_GROUPS) q = C1F_GROUPS - 1;                 // duplicate, never written
    row = (q * 2341) >> 16;                                  // q / 28 for q < 840
    xg = q - row * 28;
  };
  auto fetch_f = [&](long long j, f32x4 (&pf)[C1F_NLD][3]) {
    const int yq = (int)(j % C1_YQ);
    const int z = (int)((j / C1_YQ) % C1_D);
    const long long n = j / C1_JOBS_PER_WINDOW;
#pragma unroll
    for (int u = 0; u < C1F_NLD; ++u) {
      int row, xg;
      grp(u, row, xg);
      const int kz = row / C1_PROWS, ry = row - kz * C1_PROWS;
      const int zz = z + kz - 1, yy = yq * 2 * C1_JROWS + ry - 1;
      const bool ok = zz >= 0 && zz < C1_D && yy >= 0 && yy < C1_H;
      const float* src = p.video + (((n * C1_D + (ok ? zz : 0)) * C1_H + (ok ? yy : 0)) * (long long)C1_H + 4 * xg) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) pf[u][k] = ok ? *(const f32x4*)(src + 4 * k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park_f = [&](char* buf, const f32x4 (&pf)[C1F_NLD][3]) {
#pragma unroll
    for (int u = 0; u < C1F_NLD; ++u) {
      if (tid + 256 * u < C1F_GROUPS) {
        const float f[12] = {pf[u][0][0], pf[u][0][1], pf[u][0][2], pf[u][0][3], pf[u][1][0], pf[u][1][1],
                             pf[u][1][2], pf[u][1][3], pf[u][2][0], pf[u][2][1], pf[u][2][2], pf[u][2][3]};
        // two packed conversions per pixel (v_cvt_pk_bf16_f32): {c0, c1}, {c2, 0}
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        u32x4 w0, w1;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const f32x2_ a = {f[3 * px], f[3 * px + 1]}, b = {f[3 * px + 2], 0.f};
          const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_));
          const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_));
          if (px < 2) { w0[2 * px] = lo; w0[2 * px + 1] = hi; } else { w1[2 * (px - 2)] = lo; w1[2 * (px - 2) + 1] = hi; }
        }
        int row, xg;
        grp(u, row, xg);
        char* dst = buf + (row * WPL + 2 + 4 * xg) * 8;
        *(u32x4*)dst = w0;
        *(u32x4*)(dst + 16) = w1;
      }
    }
  };

  // per-lane fragment addressing inside a patch (bytes).  Row r of an m-tile: window w = frow>>2,
  // dy = (frow>>1)&1, dx = frow&1; K-chunk 4s+kg = taps 2(4s+kg), 2(4s+kg)+1 (taps >= 27 carry zero weights).
  const int r_w = frow >> 2, r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  const int lane_base = (r_dy * WPL + 2 * r_w + r_dx + XS) * 8;
  int toff[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int tap = 2 * (4 * s + kg) + j;
      if (tap > 26) tap = 0;
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      toff[s][j] = ((kz * C1_PROWS + ky) * WPL + kx) * 8 + lane_base;
    }

  auto load_frags = [&](const char* buf, int tile, f32x4 (&a)[2][4]) {
    const int yl = tile / C1_XG, xg = tile - yl * C1_XG;
    const char* tb = buf + (2 * yl * WPL + 16 * xg) * 8;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint2 lo = *(const uint2*)(tb + mi * 64 + toff[s][0]);
        const uint2 hi = *(const uint2*)(tb + mi * 64 + toff[s][1]);
        const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
        a[mi][s] = __builtin_bit_cast(f32x4, v);
      }
  };
  auto process = [&](long long j, int tile, const f32x4 (&a)[2][4]) {
    f32x4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[mi][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) Mma<bf16_t>::step(acc[mi][jn], a[mi][s], bfrag[s][jn]);
    bf16_t* so = s_out + wave * 8 * 72;
    unsigned char* sa = s_arg + wave * 8 * 72;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const f32x4 c = acc[mi][jn];
        const float v = fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])) + bias_v[jn], 0.f);   // pool1, bias, ReLU
        so[(mi * 4 + kg) * 72 + jn * 16 + frow] = f2bf(v);
        if (p.argmax) {
          float best = c[0];
          unsigned char idx = 0;
#pragma unroll
          for (int r = 1; r < 4; ++r)
            if (c[r] > best) { best = c[r]; idx = (unsigned char)r; }
          sa[(mi * 4 + kg) * 72 + jn * 16 + frow] = idx;
        }
      }
    __builtin_amdgcn_wave_barrier();   // wave-private LDS: DS ops of one wave execute in order
    const int px = lane >> 3, chunk = lane & 7;
    const u32x4 row = *(const u32x4*)(so + px * 72 + chunk * 8);
    const int yl = tile / C1_XG, xg = tile - yl * C1_XG;
    const int yo = (int)(j % C1_YQ) * C1_JROWS + yl;
    const int z = (int)((j / C1_YQ) % C1_D);
    const long long n = j / C1_JOBS_PER_WINDOW;
    const long long o = (((n * (C1_D + 2) + z + 1) * C1_OUT_P + yo + 1) * (long long)C1_OUT_P + xg * 8 + px + 1) * 64 + chunk * 8;
    *(u32x4*)(p.out + o) = row;
    if (p.argmax) {
      const uint2 codes = *(const uint2*)(sa + px * 72 + chunk * 8);
      const long long oa = (((n * C1_D + z) * C1_PO + yo) * (long long)C1_PO + xg * 8 + px) * 64 + chunk * 8;
      *(uint2*)(p.argmax + oa) = codes;
    }
    __builtin_amdgcn_wave_barrier();
  };

  u32x4 pf[C1_NLD];
  f32x4 pff[C1F_NLD][3];
  if constexpr (FUSED) { fetch_f(job, pff); park_f(patch, pff); }
  else { fetch(job, pf); park(patch, pf); }
  __syncthreads();
  int cur = 0;
  while (true) {
    const long long nxt = job + nslot;
    const bool more = nxt < job_end;
    if (more) {                                     // in flight while this job's 7 tiles per wave compute
      if constexpr (FUSED) fetch_f(nxt, pff); else fetch(nxt, pf);
    }
    const char* buf = patch + cur * PATCH;
    f32x4 a[2][2][4];
    load_frags(buf, wave, a[0]);
#pragma unroll
    for (int i = 0; i < C1_TILES / 4; ++i) {
      if (i + 1 < C1_TILES / 4) load_frags(buf, wave + 4 * (i + 1), a[(i + 1) & 1]);
      process(job, wave + 4 * i, a[i & 1]);
    }
    if (!more) break;
    if constexpr (FUSED) park_f(patch + (cur ^ 1) * PATCH, pff); else park(patch + (cur ^ 1) * PATCH, pf);
    __syncthreads();                                // next patch visible; everybody is done with this one
    cur ^= 1;
    job = nxt;
  }
}

}  // namespace rgp
