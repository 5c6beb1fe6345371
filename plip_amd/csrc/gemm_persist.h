// gemm_persist.h -- persistent form of the NT GEMM (same tiles, swizzle, MFMA layout and
// epilogues as gemm.h).
//
// Why: the in-kernel timeline of the one-tile-per-workgroup kernel (profiles/r01_run*_gemm_timeline.txt)
// shows a 256x256x768 tile spending ~37k cycles in its K loop, ~18k in its epilogue (all CUs store their
// tiles in the same burst while the matrix pipes idle) and ~3k waiting for its first K tile.  Here a
// workgroup stays resident and walks a strip of tiles, and the K-tile stream never stops at a tile edge:
//   * the last K iteration of tile i already pulls K tile 0 of tile i+1 into the free LDS stage, so the
//     next tile starts without a cold prologue;
//   * the epilogue only ISSUES its stores; nobody waits for them until the first barrier of the next
//     tile's K loop, one K tile of MFMA work later (vmcnt is in order on gfx9-family, so that wait also
//     covers the stores -- by then they are long gone);
//   * the per-wave transpose slabs live in the stage that was just consumed, the prefetched K tile in
//     the other one; one extra barrier per tile keeps the next stage fill off the slabs.
// Tiles are assigned statically: XCD x (workgroups with blockIdx % 8 == x) owns a contiguous strip of
// the tile space and its workgroups take tiles j, j+J, j+2J ... of that strip, so the strip's A/W panels
// stay in that XCD's L2.
#pragma once
#include "gemm.h"

namespace plipmi {

template <typename T, int BM, int BN, int WM, int WN, int EPI, int SCHED = 1>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_nt_persist_kernel(const GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int ELEMS16 = 16 / sizeof(T);
  constexpr int BK = 8 * ELEMS16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int PA = BM * 8 / NT, PW = BN * 8 / NT;
  constexpr int SLAB_PITCH = 32 * 4 + 16;          // 32 rows x 32 columns fp32 per wave, conflict-free ds_write_b128
  constexpr int SLAB_BYTES = 32 * SLAB_PITCH;
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "staging passes must be whole");
  static_assert((NT / 8) % 16 == 0, "swizzle term must not depend on the staging pass");
  static_assert(WM * WN * SLAB_BYTES <= STAGE, "epilogue slabs must fit in ONE staging buffer");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- static XCD-aware schedule ------------------------------------------------------------
  const int nbn = p.N / BN;
  const int nbm = (p.M + BM - 1) / BM;
  const int ntiles = nbm * nbn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, xj = bid >> 3, xJ = (int)gridDim.x >> 3;  // grid is a multiple of 8
  const int xq = ntiles >> 3, xr = ntiles & 7;
  const int xbase = xcd * xq + (xcd < xr ? xcd : xr);
  const int xcount = xq + (xcd < xr ? 1 : 0);
  int local = xj;
  if (local >= xcount) return;  // whole workgroup, before any barrier

  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
  const char* a_src[PA];
  const char* w_src[PW];
  auto set_tile = [&](int loc, int& m0, int& n0) {
    const int lid = xbase + loc;
    const int gw = p.gw > 0 ? p.gw : nbn, tpg = nbm * gw;
    const int cgrp = lid / tpg, crem = lid - cgrp * tpg;
    m0 = (crem / gw) * BM;
    n0 = (cgrp * gw + crem % gw) * BN;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int r = m0 + i * (NT / 8) + srow;
      r = r < p.M ? r : p.M - 1;
      a_src[i] = reinterpret_cast<const char*>(p.A) + ((size_t)r * p.lda + schunk * ELEMS16) * sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int r = n0 + i * (NT / 8) + srow;
      w_src[i] = reinterpret_cast<const char*>(p.W) + ((size_t)r * p.ldw + schunk * ELEMS16) * sizeof(T);
    }
  };
  const unsigned lds0 =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  auto stage_issue = [&](int buf) {  // LDS-DMA of the K tile a_src/w_src point at, then advance them
    const unsigned base = lds0 + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(a_src[i], base + i * NT * 16);
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16(w_src[i], base + A_BYTES + i * NT * 16);
#pragma unroll
    for (int i = 0; i < PA; ++i) a_src[i] += 128;
#pragma unroll
    for (int i = 0; i < PW; ++i) w_src[i] += 128;
  };

  const int lrow = lane & 31, lgrp = lane >> 5;
  const int lsw = (lrow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lrow * 128 + (((ks * 2 + lgrp) ^ lsw) << 4);
  const int a_tile = wm * TM * 128;
  const int w_tile = A_BYTES + wn * TN * 128;

  f32x16 acc[MI][NI];
  auto compute = [&](int buf) {
    const char* sb = smem + buf * STAGE;
    u32x4 xf[2][MI], wf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[0][i] = *reinterpret_cast<const u32x4*>(sb + a_tile + i * 32 * 128 + foff[0]);
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[0][j] = *reinterpret_cast<const u32x4*>(sb + w_tile + j * 32 * 128 + foff[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          xf[(ks + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sb + a_tile + i * 32 * 128 + foff[ks + 1]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          wf[(ks + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sb + w_tile + j * 32 * 128 + foff[ks + 1]);
      }
      if constexpr (SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) mma16<T>(acc[i][j], wf[ks & 1][j], xf[ks & 1][i]);
      if constexpr (SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int KT = p.K / BK;
  int m0, n0;
  set_tile(local, m0, n0);
  int st = 0;  // stage holding the K tile about to be multiplied
  stage_issue(st);
  wait_vm0();
  __syncthreads();

  for (;;) {
    unsigned long long* trace = p.trace ? p.trace + (size_t)(xbase + local) * 8 : nullptr;
    if (trace && tid == 0) {
      trace[0] = __builtin_amdgcn_s_memtime();
      trace[1] = trace[0];
      trace[4] = xbase + local;
      trace[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                 ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
      trace[6] = KT;
      trace[7] = bid;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nlocal = local + xJ;
    const bool more = nlocal < xcount;
    int nm0 = 0, nn0 = 0;
    // ---- K loop; every iteration prefetches the NEXT K tile of the stream, across the tile edge
    for (int kt = 0; kt < KT; ++kt) {
      const bool last = kt + 1 == KT;
      const bool fetch = !last || more;
      if (last && more) set_tile(nlocal, nm0, nn0);  // this tile's loads are all issued: retarget the pointers
      if (fetch) stage_issue(st ^ 1);
      compute(st);
      if (fetch) wait_vm0();
      __syncthreads();
      st ^= 1;
    }
    if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();

    // ---- epilogue: slabs in the stage just consumed (st^1); `st` holds the next tile's K tile 0 ----
    {
      char* slab = smem + (st ^ 1) * STAGE + wave * SLAB_BYTES;
      const int rd_row = lane >> 3, rd_col = (lane & 7) * 4;  // 8 lanes x 16 B = one 128-byte slab row
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        // one memory latency per 32-row block: all bias / residual / position rows requested up front
        float4 add[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = m0 + wm * TM + i * 32 + it * 8 + rd_row;
            add[j][it] = EpilogueOp<T, EPI>::load(p, m < p.M ? m : p.M - 1, n0 + wn * TN + j * 32 + rd_col);
          }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(slab + lrow * SLAB_PITCH + (8 * q + 4 * lgrp) * 4) = v;
          }
          __builtin_amdgcn_wave_barrier();
          const int n = n0 + wn * TN + j * 32 + rd_col;
          f32x4 v[4];
#pragma unroll
          for (int it = 0; it < 4; ++it)
            v[it] = *reinterpret_cast<const f32x4*>(slab + (it * 8 + rd_row) * SLAB_PITCH + rd_col * 4);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = m0 + wm * TM + i * 32 + it * 8 + rd_row;
            if (m < p.M) EpilogueOp<T, EPI>::store(p, m, n, v[it][0], v[it][1], v[it][2], v[it][3], add[j][it]);
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (trace && tid == 0) trace[3] = __builtin_amdgcn_s_memtime();  // stores issued, not necessarily retired
    if (!more) break;
    __syncthreads();  // every wave is off its slab before the next K loop refills that stage
    local = nlocal;
    m0 = nm0;
    n0 = nn0;
  }
}

}  // namespace plipmi
