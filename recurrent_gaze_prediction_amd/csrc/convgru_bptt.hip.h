// Persistent BPTT of the ConvGRU (gfx950, bf16 operands): all T backward steps of the recurrence of
// /root/reference/models/gaze_grcn.py:95-129 (what tf.gradients builds for the unrolled cell, base.py:278-281) in ONE
// launch -- the mirror of convgru_seq.hip.h.  The per-step path (rgp_grcn_bwd.hip) runs 4 T dependent launches.
//
// Per step t (descending), for the gradient dh arriving at h_t (head part dh_head[t] + carry from step t+1):
//   du = dh (h_{t-1} - c), dc = dh (1 - u), carry = dh u;   dz_pre = du u (1-u),  dc_pre = dc (1 - c^2)
//   d(r.h) = conv3x3(dc_pre ; rot180 U^T)                    dr_pre = d(r.h) h_{t-1} r (1-r),  carry += d(r.h) r
//   carry += conv3x3([dz_pre | dr_pre] ; rot180 [U_z ; U_r]^T)
// and dXpre[:, t] = [dz_pre | dr_pre | dc_pre] is what the hoisted filter gradients consume afterwards.
//
// Same decomposition as the forward: a group of 8 workgroups owns up to 2 clips, member j the 16 state channels
// [16j, 16j+16) of every quantity above; its columns of the two transposed filters (K = 1152 and 2304: 27 k-steps per
// wave and K-quarter = 108 VGPRs) stay in registers for the whole sequence, the carry of a tile in the registers of
// the wave that finalises it.  dc_pre, then dz_pre | dr_pre (bf16 operand images) are exchanged between the members
// twice per step with the write-through / phase-counter protocol of convgru_seq.hip.h.
#pragma once
#include "convgru_seq.hip.h"

namespace rgp {

struct BpttParams {
  const bf16_t* w_c;         // packed dgrad filter of U:        [128][K = tap*128 + o]
  const bf16_t* w_zr;        // packed dgrad filter of U_z|U_r:  [128][K = tap*256 + gate*128 + o]
  const float* dh_head;      // [T][B][49][128] gradient reaching h_t from the head (after the batch-norm backward)
  const float* hall;         // [T+1][B][49][128]
  const float* uall;         // [T][B][49][128]
  const float* rall;
  const float* call;
  float* dxpre;              // [B][T][49][384]
  bf16_t* xch_c;             // [ngroups][98][128] exchange images
  bf16_t* xch_z;
  bf16_t* xch_r;
  unsigned* cnt;             // [ngroups][2T] phase counters, zeroed before the launch
  unsigned* err;             // host-visible error word of the plan (convgru_seq.hip.h)
  int B, T, NC, ngroups;
  int skip_member;           // fault injection: this member of group 0 leaves at once; -1 = none
};

template <int NF>
static __global__ __launch_bounds__(SEQ_NT) void convgru_bptt_kernel(const BpttParams p) {
  extern __shared__ __attribute__((aligned(16))) char sq_smem[];
  char* img_a = sq_smem;                 // dc_pre, later dz_pre
  char* img_b = sq_smem + SEQ_IMG;       // dr_pre
  char* red = sq_smem + SEQ_RED_OFF;     // 4 x NF partial tiles of 1 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* stage = sq_smem + SEQ_STAGE_OFF + kq * 512;

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

  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bc[9], bzr[18];
  {
    const bf16_t* wc = p.w_c + (long long)(16 * j + frow) * (9 * S);
    const bf16_t* wz = p.w_zr + (long long)(16 * j + frow) * (18 * S);
#pragma unroll
    for (int i = 0; i < 9; ++i) bc[i] = *(const f32x4*)(wc + (kq * 9 + i) * 32 + fk * 8);
#pragma unroll
    for (int i = 0; i < 18; ++i) bzr[i] = *(const f32x4*)(wz + (kq * 18 + i) * 32 + fk * 8);
  }
  int abase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int m = f * 16 + frow;
    int pix = 2 * 81;
    if (m < rows) { const int c = m / 49, q = m - c * 49; pix = c * 81 + (q / 7) * 9 + (q % 7); }
    abase[f] = pix * SEQ_PIXB + fk * 16;
  }
  int orow[2][4];
  bool ovalid[2][4];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      orow[o][r] = (kq + 4 * o) * 16 + fk * 4 + r;
      ovalid[o][r] = (kq + 4 * o) < NF && orow[o][r] < rows;
    }
  const int ch = 16 * j + frow;
  float carry[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const unsigned xbytes = (unsigned)p.ngroups * 98u * 256u;
  unsigned* cnt = p.cnt + (long long)group * 2 * T_;
  int& s_timeout = *(int*)(sq_smem + SEQ_FLAG_OFF);
  if (tid == 0) s_timeout = 0;
  __syncthreads();

  auto mma = [&](const f32x4 (&a)[NF], const f32x4& b, f32x4 (&acc)[NF]) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a[f]), __builtin_bit_cast(s16x8, b), acc[f], 0, 0, 0);
  };
  auto publish_tile = [&](bf16_t* xch, int f, const float (&v)[4]) {
    bf16_t* sg = (bf16_t*)stage;
#pragma unroll
    for (int r = 0; r < 4; ++r) sg[(fk * 4 + r) * 16 + frow] = f2bf(v[r]);
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
  auto arrive_wait = [&](int ph) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt + ph, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // bounded: ~1 s; a workgroup that timed out once stops waiting altogether (its results are poisoned below), so a
      // group with a missing member costs a second, not a second per phase
      if (!s_timeout && !seq_wait_phase(cnt + ph)) s_timeout = 1;
    }
    __syncthreads();
  };
  auto load_image = [&](const bf16_t* xch, char* img) {
    for (int i = tid; i < rows * 16; i += SEQ_NT) {
      const int row = i >> 4, c16 = i & 15;
      const u32x4 q = seq_ld_sc1(xch, xbytes, (unsigned)(((group * 98 + row) * 128 + c16 * 8) * 2));
      const int c = row / 49, r49 = row - c * 49;
      const int pix = c * 81 + (r49 / 7 + 1) * 9 + (r49 % 7 + 1);
      *(u32x4*)(img + pix * SEQ_PIXB + c16 * 16) = q;
    }
  };
  // sum of the 4 K-quarter partials of an owned tile
  auto reduce_tile = [&](int f) {
    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) sacc += *(const f32x4*)(red + ((q * NF + f) << 10) + lane * 16);
    return sacc;
  };

  for (int t = T_ - 1; t >= 0; --t) {
    // saved forward quantities of this lane's rows (own channels)
    float hp[2][4], uu[2][4], rr[2][4], cc[2][4], dh[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hp[o][r] = uu[o][r] = rr[o][r] = cc[o][r] = dh[o][r] = 0.f;
        if (ovalid[o][r]) {
          const long long off = (long long)t * st + ((long long)(clip0 * 49 + orow[o][r])) * S + ch;
          hp[o][r] = p.hall[off]; uu[o][r] = p.uall[off]; rr[o][r] = p.rall[off]; cc[o][r] = p.call[off];
          dh[o][r] = p.dh_head[off] + carry[o][r];
        }
      }
    float dzp[2][4], dcp[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float du = dh[o][r] * (hp[o][r] - cc[o][r]), dc = dh[o][r] * (1.f - uu[o][r]);
        carry[o][r] = dh[o][r] * uu[o][r];
        dcp[o][r] = dc * (1.f - cc[o][r] * cc[o][r]);
        dzp[o][r] = du * uu[o][r] * (1.f - uu[o][r]);
      }
#pragma unroll
    for (int o = 0; o < 2; ++o)
      if (kq + 4 * o < NF) publish_tile(p.xch_c, kq + 4 * o, dcp[o]);
    arrive_wait(2 * (T_ - 1 - t));
    load_image(p.xch_c, img_a);
    __syncthreads();
    // the plain outputs go out BEHIND the hand-off, under the MFMAs that follow (convgru_seq.hip.h): in front of it the
    // hand-off's drain waits for their HBM acknowledgements
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ovalid[o][r]) {
          const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
          float* row = p.dxpre + (((long long)(clip0 + c) * T_ + t) * 49 + r49) * (3 * S) + ch;
          row[0] = dzp[o][r];
          row[2 * S] = dcp[o][r];
        }

    // ---- d(r.h) = conv3x3(dc_pre; U^T): this wave's K quarter, reduced through LDS
    {
      f32x4 acc[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int ks = kq * 9 + i, tap = ks >> 2, cb = ks & 3;
        const int toff = ((tap / 3) * 9 + tap % 3) * SEQ_PIXB + cb * 64;
        f32x4 a[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) a[f] = *(const f32x4*)(img_a + abase[f] + toff);
        mma(a, bc[i], acc);
        if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) *(f32x4*)(red + ((kq * NF + f) << 10) + lane * 16) = acc[f];
    }
    __syncthreads();
    float drp[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (kq + 4 * o < NF) d = reduce_tile(kq + 4 * o);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        drp[o][r] = d[r] * hp[o][r] * rr[o][r] * (1.f - rr[o][r]);
        carry[o][r] += d[r] * rr[o][r];
      }
    }
    auto store_drp = [&]() {
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ovalid[o][r]) {
            const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
            p.dxpre[(((long long)(clip0 + c) * T_ + t) * 49 + r49) * (3 * S) + S + ch] = drp[o][r];
          }
    };
    // (the last step's carry is not needed by anybody: the recurrence starts from h_0 = 0, gaze_grcn.py:262)
    if (t == 0) { store_drp(); break; }
#pragma unroll
    for (int o = 0; o < 2; ++o)
      if (kq + 4 * o < NF) { publish_tile(p.xch_z, kq + 4 * o, dzp[o]); publish_tile(p.xch_r, kq + 4 * o, drp[o]); }
    arrive_wait(2 * (T_ - 1 - t) + 1);                    // (its barriers also order the partial-tile reads above)
    load_image(p.xch_z, img_a);
    load_image(p.xch_r, img_b);
    __syncthreads();
    store_drp();                                           // behind the hand-off, as above

    // ---- carry += conv3x3([dz_pre | dr_pre]; [U_z ; U_r]^T)
    {
      f32x4 acc[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        const int ks = kq * 18 + i, tap = ks >> 3, gate = (ks >> 2) & 1, cb = ks & 3;
        const char* img = gate ? img_b : img_a;
        const int toff = ((tap / 3) * 9 + tap % 3) * SEQ_PIXB + cb * 64;
        f32x4 a[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) a[f] = *(const f32x4*)(img + abase[f] + toff);
        mma(a, bzr[i], acc);
        if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) *(f32x4*)(red + ((kq * NF + f) << 10) + lane * 16) = acc[f];
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (kq + 4 * o < NF) {
        const f32x4 d = reduce_tile(kq + 4 * o);
#pragma unroll
        for (int r = 0; r < 4; ++r) carry[o][r] += d[r];
      }
    }
    __syncthreads();                                      // partial tiles read before the next step overwrites them
  }
  // a group that timed out must not look like a result
  if (s_timeout) {
    if (tid == 0 && p.err) { *(volatile unsigned*)p.err = 1u; __threadfence_system(); }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ovalid[o][r]) {
          const int c = orow[o][r] / 49, r49 = orow[o][r] - c * 49;
          p.dxpre[(((long long)(clip0 + c) * T_) * 49 + r49) * (3 * S) + ch] = __builtin_nanf("");
        }
  }
}

}  // namespace rgp
