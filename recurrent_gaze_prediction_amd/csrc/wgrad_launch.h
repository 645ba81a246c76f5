// Host-side launch of wgrad_kernel: tiles of 4 K-chunks x 128 (WNT 4) or 256 (WNT 8) output channels, the row
// reduction split over blockIdx.y so that about a thousand blocks are in flight.
#pragma once
#include <algorithm>

#include "rgp_host.h"
#include "wgrad.hip.h"

namespace rgp {

template <typename T, int G, int WNT>
int launch_wgrad_t(const WgradParams& p, hipStream_t s) {
  auto kern = wgrad_kernel<T, G, WNT>;
  constexpr int smem = WgradSmem<T, WNT>::BYTES;
  constexpr int BN = 32 * WNT;
  RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
  if (p.M <= 0 || p.nk <= 0 || p.N <= 0) return set_err(RGP_EINVAL, "wgrad: empty problem");
  const int n_kt = (p.nk + 3) / 4, n_nt = (p.N + BN - 1) / BN;
  const long long total_steps = (p.M + 31) / 32;
  // row ranges (splits): enough for ~1000 blocks (RGP_WGK_BLOCKS overrides), at least 8 -- several rounds of blocks even
  // out the tiles' different lengths.  Every block ends with one fp32 atomic per element of its 256 x BN tile, so a
  // problem whose blocks would then run fewer than 256 steps (the head's and the recurrent cells' filter gradients:
  // M = B T 49 rows) gets ONE round of blocks instead (2 per CU for the 128-wide tile, 1 for the 256-wide one): its
  // atomic traffic, not its MFMAs, is what takes the time (cfg-4 training step 2.79 -> 2.38 ms, B 64 x T 16 4.23 -> 3.81).
  const int target = dev_knob("RGP_WGK_BLOCKS", 1024);
  const int nz = std::max(1, std::min(p.nz, 8));
  const int tiles = n_kt * n_nt * nz;
  long long splits = std::max<long long>(8, target / tiles);
  splits = std::min(splits, total_steps);
  if (total_steps / splits < 256) {
    int n_cu = 256;
    RGP_TRY(device_cu_count(&n_cu));
    const int slots = (WNT == 4 && sizeof(T) == 2 ? 2 : 1) * n_cu;
    splits = std::max<long long>(1, std::min<long long>(slots / tiles, total_steps / 16));
    splits = std::max<long long>(1, std::min(splits, total_steps));
  }
  WgradParams q = p;
  {
    const WgradGeom g = {p.D, p.H, p.W, p.x_sz, p.x_sy, p.x_sx, p.y_sz, p.y_sy, p.y_sx, p.y_org, (int)sizeof(T), p.x_img_stride, p.y_img_stride};
    RGP_TRY(wgrad_row_tables(g, s, &q.x_tab, &q.y_tab));
  }
  q.ablate = dev_knob("RGP_WG_ABLATE", 0);
  q.steps_per_split = (int)((total_steps + splits - 1) / splits);
  splits = (total_steps + q.steps_per_split - 1) / q.steps_per_split;
  kern<<<dim3(n_kt * n_nt, (unsigned)splits, (unsigned)nz), 512, smem, s>>>(q);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// The 256-wide tile (one block per CU, a third fewer LDS-DMA bytes per FLOP) pays only for few tiles with a long
// reduction -- conv3a / conv3b: -3 % -- and loses 5...15 % where the 128-wide kernel already has hundreds of
// tiles (conv4*, conv5*: two co-resident blocks hide each other's barriers).  RGP_WG_WIDE=0 / 2 forces never / always.
// (A variant with the two wave groups of the 256-wide tile running half a step apart, as in igemm_stagger.hip.h, measured
// no gain on the fine-tune step and was removed.)
template <typename T, int G>
int launch_wgrad(const WgradParams& p, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const int wide = dev_knob("RGP_WG_WIDE", 1);
    const int narrow_tiles = ((p.nk + 3) / 4) * ((p.N + 127) / 128);
    if (wide && p.N % 256 == 0 && (wide == 2 || narrow_tiles <= 64)) return launch_wgrad_t<T, G, 8>(p, s);
  }
  return launch_wgrad_t<T, G, 4>(p, s);
}

// geometry helpers: rows of one image are the positions (z, y, x) of a D x H x W grid
inline void wgrad_grid(WgradParams& p, int D, int H, int W) {
  p.D = D; p.H = H; p.W = W;
  p.Mw = D * H * W;
}

}  // namespace rgp
