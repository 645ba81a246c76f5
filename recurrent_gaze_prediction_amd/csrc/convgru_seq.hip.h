// Persistent ConvGRU sequence kernel (gfx950, bf16 operands): ALL T steps of GRU_RCN_Cell.__call__
// (/root/reference/models/gaze_grcn.py:95-129, unrolled at :259-288) plus the per-timestep inference batch-norm
// (:325) in ONE launch.  The per-step path (rgp_grcn.hip seq_impl) runs 2 T dependent launches of ~13 us whose
// M = B*49 rows fill a fraction of the chip; here the recurrent filters never move and only the state does.
//
// Decomposition.  The serial critical path is  h_{t-1} -> [U_z|U_r] conv -> r.h -> U conv -> h_t , 43 MFLOP per clip
// and step against 884 KB of bf16 filters: re-streaming the filters per step costs 12.6 us per CU at the measured
// 66-73 GB/s L2 -> LDS ingest, however the clips are dealt.  So the filters are made RESIDENT: a group of 8 CUs
// (workgroups) owns up to 2 clips; member j keeps the z, r and c filter columns of state channels [16j, 16j+16) in
// REGISTERS (3 x 16 columns x K = 1152: 110 KB per workgroup = 108 VGPRs per lane, loaded once), split over its
// 4 waves (one per SIMD, 512 registers each) as 4 K-quarters (9 MFMA k-steps of 32 each).  Per step a member computes its 16 channels of
//   u = sigmoid(W_z*x + U_z*h), r = sigmoid(W_r*x + U_r*h)        (x-parts hoisted: xpre)
//   c = tanh(W*x + U*(r.h)),  h' = u.h + (1-u).c
// for the group's clips; r.h and h' (bf16 operand images, 12.5 KB per clip) are exchanged between the 8 members
// through an L2-resident buffer twice per step.  The fp32 state of a tile lives in the registers of the wave that
// finalises it for the whole sequence.
//
// Exchange protocol (cdna_hip_programming.md Guideline 16 R1, counter form; placement-independent): every payload byte
// is stored write-through (sc1) as 16-byte rows, every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup
// barriers, ONE lane adds to the group's monotonic phase counter; consumers poll that counter with sc1 loads from one
// lane (bounded, with s_sleep), barrier, and read the payload with sc1 loads only.  Counters are zeroed by a
// hipMemsetAsync ahead of the launch.  All 8 x ngroups <= 256 workgroups (256 threads, 157 KB of LDS: one per CU) must
// be resident together; the host checks the CU count and falls back to per-step launches otherwise.  A member that
// never arrives (e.g. two such launches interleaved on one device from different streams: keep ONE in flight per
// device) makes its group time out after ~1 s, NaN-poison its final state and leave -- a loud failure, never a hang.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct SeqParams {
  const bf16_t* w_zr;        // packed [256][K] (rows 0..127 U_z columns, 128..255 U_r), K = tap*128 + c
  const bf16_t* w_c;         // packed [128][K]
  const float* xpre;         // [B][T][49][384] hoisted W_z|W_r|W * x
  float* hall;               // [T+1][B][49][128] fp32 states (slot 0 = h_0 = 0, pre-zeroed)
  float* uall;               // [T][B][49][128]
  float* rall;               // optional (training)
  float* call;               // optional (training)
  bf16_t* hbn;               // [B*T][81][128] halo-padded BN(h_t): the head's input image of frame b*T+t
  const float* bn_gamma;     // [T][128]
  const float* bn_beta;
  float bn_inv_std;
  bf16_t* xch_h;             // [ngroups][98][128] exchange image of h'
  bf16_t* xch_rh;            // [ngroups][98][128] exchange image of r.h
  unsigned* cnt;             // [ngroups][2T] phase counters, zeroed before the launch
  unsigned* err;             // host-visible error word of the plan (pinned, mapped): set to 1 by a group that timed out
  int B, T, NC, ngroups, K;
  int skip_member;           // fault injection (rgp_grcn_inject_fault): this member of group 0 leaves at once; -1 = none
};

constexpr int SEQ_PIXB = 272;                        // bytes per padded pixel: 128 ch bf16 + 16 pad (bank rotation)
constexpr int SEQ_NPIX = 2 * 81 + 24;                // two 9x9 images + a zero region for padding rows (all 9 taps)
constexpr int SEQ_IMG = SEQ_NPIX * SEQ_PIXB;         // 50 592 B
constexpr int SEQ_RED_OFF = 2 * SEQ_IMG;             // 4 waves x 7 fragments x 2 gates partial tiles of 1 KiB (56 KiB)
constexpr int SEQ_STAGE_OFF = SEQ_RED_OFF + 56 * 1024;
constexpr int SEQ_FLAG_OFF = SEQ_STAGE_OFF + 4 * 512;
constexpr int SEQ_SMEM = SEQ_FLAG_OFF + 16;          // 160 592 B (no static __shared__: the dynamic base stays 16-B aligned)
static_assert(SEQ_SMEM <= 160 * 1024, "LDS budget");

__device__ __forceinline__ u32x4 seq_ld_sc1(const void* base, unsigned bytes, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), off, 0, 16));
}
__device__ __forceinline__ void seq_st_sc1(void* base, unsigned bytes, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000), off, 0, 16);
}

constexpr int SEQ_NT = 256;                          // 4 waves = one per SIMD, each with the whole 512-register file

// Bounded wait for a group's phase counter (one lane).  The bound is a DEADLINE on the constant-rate real-time counter
// (s_memrealtime: 100 MHz on gfx950, independent of the shader clock), not an iteration count: how long an iteration takes
// depends on the clock the chip holds and on what else loads the same L2 channel (e.g. a concurrent RCCL kernel), so a
// count (rounds 2-4: 2^20 iterations) bounds nothing in particular.  ~1 s of wall clock; the clock is read every 64th poll.
constexpr unsigned long long SEQ_DEADLINE_TICKS = 100ull * 1000 * 1000;
__device__ __forceinline__ bool seq_wait_phase(const unsigned* cnt) {      // true = all 8 members arrived, false = deadline passed
  if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 8u) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned spins = 1;; ++spins) {
    __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 8u) return true;
    if ((spins & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > SEQ_DEADLINE_TICKS) return false;
  }
}

// NF = 16-row fragments of a group: 4 (one clip, 49 rows) or 7 (two clips, 98 rows); a template parameter so that the
// MFMA loops carry no run-time guards (measured: wave-uniform `if (f < MF)` around the reads / MFMAs cost 25 %).
template <int NF>
static __global__ __launch_bounds__(SEQ_NT) void convgru_seq_kernel(const SeqParams p) {
  extern __shared__ __attribute__((aligned(16))) char sq_smem[];
  char* img_h = sq_smem;
  char* img_rh = sq_smem + SEQ_IMG;
  char* red_zr = sq_smem + SEQ_RED_OFF;  // 4 x 7 x 2 partial tiles of 1 KiB, z|r phase.  NOT overlaid on the images: their
  char* red_c = red_zr;                  // halo pixels and zero region must stay zero for the whole sequence.
  const int tid = threadIdx.x, lane = tid & 63;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = K quarter
  char* stage = sq_smem + SEQ_STAGE_OFF + kq * 512;

  // group / member of this workgroup; with a multiple of 8 groups a group's members sit on one XCD (speed only)
  int group, j;
  {
    const int b = blockIdx.x;
    if ((p.ngroups & 7) == 0) { const int slot = b >> 3; group = (slot >> 3) * 8 + (b & 7); j = slot & 7; }
    else { group = b >> 3; j = b & 7; }
  }
  if (group == 0 && j == p.skip_member) return;           // fault injection: a member that never arrives
  const int clip0 = group * p.NC;
  const int nclip = min(p.NC, p.B - clip0);
  const int rows = nclip * 49;
  const int S = 128, T_ = p.T;
  const long long st = (long long)p.B * 49 * S;

  for (int i = tid; i < (2 * SEQ_IMG) / 16; i += SEQ_NT) ((u32x4*)sq_smem)[i] = (u32x4){0u, 0u, 0u, 0u};

  // ---- resident filter fragments: k-steps [9 kq, 9 kq + 9) of the z, r, c columns of channels 16 j .. 16 j + 15
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bz[9], br[9], bc[9];
  {
    const bf16_t* wz = p.w_zr + (long long)(16 * j + frow) * p.K;
    const bf16_t* wr = p.w_zr + (long long)(128 + 16 * j + frow) * p.K;
    const bf16_t* wc = p.w_c + (long long)(16 * j + frow) * p.K;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = (kq * 9 + i) * 32 + fk * 8;
      bz[i] = *(const f32x4*)(wz + k);
      br[i] = *(const f32x4*)(wr + k);
      bc[i] = *(const f32x4*)(wc + k);
    }
  }
  // per-lane A-fragment bases: row lane & 15 of fragment f -> pixel of tap (0,0); padding rows -> the zero region
  int abase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int m = f * 16 + frow;
    int pix = 2 * 81;
    if (m < rows) { const int c = m / 49, q = m - c * 49; pix = c * 81 + (q / 7) * 9 + (q % 7); }
    abase[f] = pix * SEQ_PIXB + fk * 16;
  }
  // this wave finalises fragments kq and kq + 4; a lane holds 4 rows of each tile (accumulator layout: row 4 (lane >> 4)
  // + r, column lane & 15) and keeps their fp32 state for the whole sequence
  int orow[2][4];
  bool ovalid[2][4];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      orow[o][r] = (kq + 4 * o) * 16 + fk * 4 + r;
      ovalid[o][r] = orow[o][r] < rows;
    }
  const int ch = 16 * j + frow;                          // this lane's state channel
  float h_prev[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const unsigned xbytes = (unsigned)p.ngroups * 98u * 256u;
  unsigned* cnt = p.cnt + (long long)group * 2 * T_;
  int& s_timeout = *(int*)(sq_smem + SEQ_FLAG_OFF);
  if (tid == 0) s_timeout = 0;
  __syncthreads();

  // A fragments of k-step i of this wave's K quarter, and the MFMAs of one filter column block on them
  auto a_frags = [&](const char* img, int i, f32x4 (&a)[NF]) {
    const int ks = kq * 9 + i, tap = ks >> 2, cb = ks & 3;
    const int toff = ((tap / 3) * 9 + tap % 3) * SEQ_PIXB + cb * 64;
#pragma unroll
    for (int f = 0; f < NF; ++f) a[f] = *(const f32x4*)(img + abase[f] + toff);
  };
  auto mma7 = [&](const f32x4 (&a)[NF], const f32x4& b, f32x4 (&acc)[NF]) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a[f]), __builtin_bit_cast(s16x8, b), acc[f], 0, 0, 0);
  };
  // publish an owned 16 x 16 tile (bf16) as 16-byte rows into an exchange image, write-through
  auto publish_tile = [&](bf16_t* xch, int f, const float (&v)[4]) {
    bf16_t* sg = (bf16_t*)stage;
#pragma unroll
    for (int r = 0; r < 4; ++r) sg[(fk * 4 + r) * 16 + frow] = f2bf(v[r]);
    // DS operations of one wave execute in order; the COMPILER must not move the 16-byte reads above the 2-byte
    // stores (different access types: type-based alias analysis would let it)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (lane < 32) {
      const int row = f * 16 + (lane >> 1);
      if (row < rows) {
        const u32x4 q = *(const u32x4*)(stage + lane * 16);
        seq_st_sc1(xch, xbytes, (unsigned)(((group * 98 + row) * 128 + 16 * j + (lane & 1) * 8) * 2), q);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };
  // group rendezvous, part 1: this member's tiles of phase `ph` are published (drain, barrier, one counter add)
  auto arrive = [&](int ph) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt + ph, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // part 2: all 8 members have published; load the exchange image into an LDS image.  (The step's plain output stores
  // are issued after it, see the loop.  Measured and not kept: Guideline 16's R2 form -- 8-byte {tag, 2 x bf16} granules
  // polled by every member instead of image + counter: 0.61 ms instead of 0.23 at B = 64, T = 16; 25 atomic loads per
  // thread and sweep over a 50 KB image are far beyond the <= 4 KB that form is meant for.)
  auto wait_load = [&](int ph, const bf16_t* xch, char* img) {
    if (tid == 0) {
      // bounded: ~1 s; a workgroup that timed out once stops waiting altogether (its results are poisoned below), so a
      // group with a missing member costs a second, not a second per phase
      if (!s_timeout && !seq_wait_phase(cnt + ph)) s_timeout = 1;
    }
    __syncthreads();
    for (int i = tid; i < rows * 16; i += SEQ_NT) {
      const int row = i >> 4, c16 = i & 15;
      const u32x4 q = seq_ld_sc1(xch, xbytes, (unsigned)(((group * 98 + row) * 128 + c16 * 8) * 2));
      const int c = row / 49, r49 = row - c * 49;
      const int pix = c * 81 + (r49 / 7 + 1) * 9 + (r49 % 7 + 1);
      *(u32x4*)(img + pix * SEQ_PIXB + c16 * 16) = q;
    }
    __syncthreads();
  };

  for (int t = 0; t < T_; ++t) {
    // hoisted input parts of this lane's rows x {z, r, c} (in flight during the MFMAs)
    float xz[2][4], xr[2][4], xc[2][4], gam, bet;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xz[o][r] = xr[o][r] = xc[o][r] = 0.f;
        if (ovalid[o][r]) {
          const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
          const float* xp = p.xpre + (((long long)(clip0 + c) * T_ + t) * 49 + r49) * (3 * S) + ch;
          xz[o][r] = xp[0]; xr[o][r] = xp[S]; xc[o][r] = xp[2 * S];
        }
      }
    gam = p.bn_gamma[t * S + ch];
    bet = p.bn_beta[t * S + ch];

    // ---- z | r phase: partial sums of this wave's K quarter, reduced over the 4 quarters through LDS
    {
      f32x4 az[NF], ar[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) { az[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; ar[f] = az[f]; }
      f32x4 a0[NF], a1[NF];                              // two k-steps of A fragments in flight (software pipeline)
      a_frags(img_h, 0, a0);
#pragma unroll
      for (int i = 0; i < 9; i += 2) {
        if (i + 1 < 9) a_frags(img_h, i + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mma7(a0, bz[i], az); mma7(a0, br[i], ar);
        if (i + 2 < 9) a_frags(img_h, i + 2, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < 9) { mma7(a1, bz[i + 1], az); mma7(a1, br[i + 1], ar); }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        *(f32x4*)(red_zr + (((kq * NF + f) * 2 + 0) << 10) + lane * 16) = az[f];
        *(f32x4*)(red_zr + (((kq * NF + f) * 2 + 1) << 10) + lane * 16) = ar[f];
      }
    }
    __syncthreads();
    float u[2][4], rh[2][4], rgs[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int f = kq + 4 * o;
      f32x4 sz = (f32x4){0.f, 0.f, 0.f, 0.f}, sr = sz;
      if (f < NF) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sz += *(const f32x4*)(red_zr + (((q * NF + f) * 2 + 0) << 10) + lane * 16);
          sr += *(const f32x4*)(red_zr + (((q * NF + f) * 2 + 1) << 10) + lane * 16);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        u[o][r] = sigmoidf_(sz[r] + xz[o][r]);
        rgs[o][r] = sigmoidf_(sr[r] + xr[o][r]);
        rh[o][r] = rgs[o][r] * h_prev[o][r];
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
      if (kq + 4 * o < NF) publish_tile(p.xch_rh, kq + 4 * o, rh[o]);
    arrive(2 * t);
    wait_load(2 * t, p.xch_rh, img_rh);
    // this phase's plain outputs go out BEHIND the hand-off, under the candidate phase's MFMAs: in front of it they sit
    // in the queue that arrive() drains (the hand-off then waits for their HBM acknowledgements), between its two parts
    // they delay the poll and the image loads (+25 % per step, measured)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ovalid[o][r]) {
          const long long off = (long long)t * st + ((long long)(clip0 * 49 + orow[o][r])) * S + ch;
          p.uall[off] = u[o][r];
          if (p.rall) p.rall[off] = rgs[o][r];
        }

    // ---- candidate phase on r.h, blend, batch-norm
    {
      f32x4 ac[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) ac[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 a0[NF], a1[NF];
      a_frags(img_rh, 0, a0);
#pragma unroll
      for (int i = 0; i < 9; i += 2) {
        if (i + 1 < 9) a_frags(img_rh, i + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mma7(a0, bc[i], ac);
        if (i + 2 < 9) a_frags(img_rh, i + 2, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < 9) mma7(a1, bc[i + 1], ac);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) *(f32x4*)(red_c + ((kq * NF + f) << 10) + lane * 16) = ac[f];
    }
    __syncthreads();
    float hn[2][4], cgs[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int f = kq + 4 * o;
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (f < NF) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sc += *(const f32x4*)(red_c + ((q * NF + f) << 10) + lane * 16);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cgs[o][r] = tanhf_(sc[r] + xc[o][r]);
        hn[o][r] = u[o][r] * h_prev[o][r] + (1.f - u[o][r]) * cgs[o][r];
        h_prev[o][r] = hn[o][r];
      }
    }
    if (t + 1 < T_) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
        if (kq + 4 * o < NF) publish_tile(p.xch_h, kq + 4 * o, hn[o]);
      arrive(2 * t + 1);
      wait_load(2 * t + 1, p.xch_h, img_h);
    }
    // (outputs behind the hand-off, as above: they drain under the next step's z | r MFMAs)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ovalid[o][r]) {
          const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
          const long long off = ((long long)(clip0 * 49 + orow[o][r])) * S + ch;
          if (p.call) p.call[(long long)t * st + off] = cgs[o][r];
          p.hall[(long long)(t + 1) * st + off] = hn[o][r];
          const long long fr = (long long)(clip0 + c) * T_ + t;
          p.hbn[(fr * 81 + (r49 / 7 + 1) * 9 + (r49 % 7 + 1)) * S + ch] = f2bf(gam * (hn[o][r] * p.bn_inv_std) + bet);
        }
  }
  // a group that timed out must not look like a result: the head reads hbn (every frame of the group's clips: the
  // exchange images were stale from the first missed phase on), the training path hall
  if (s_timeout) {
    if (tid == 0 && p.err) { *(volatile unsigned*)p.err = 1u; __threadfence_system(); }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ovalid[o][r]) {
          const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
          const long long off = ((long long)(clip0 * 49 + orow[o][r])) * S + ch;
          for (int t = 0; t < T_; ++t) {
            const long long fr = (long long)(clip0 + c) * T_ + t;
            p.hbn[(fr * 81 + (r49 / 7 + 1) * 9 + (r49 % 7 + 1)) * S + ch] = (bf16_t)0x7FC0;     // bf16 NaN
            p.hall[(long long)(t + 1) * st + off] = __builtin_nanf("");
          }
        }
  }
}

}  // namespace rgp
