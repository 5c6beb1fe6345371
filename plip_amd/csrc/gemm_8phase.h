// gemm_8phase.h -- 256x256x(BK) NT GEMM with a half-tile LDS ring and two wave groups in antiphase.
//
// Same operand layout as gemm.h (128-byte LDS rows, XOR swizzle, swapped 32x32 MFMA operands, slab
// epilogue); what changes is the schedule of the K loop.  Measurements on the one-barrier-per-K-tile
// kernels (profiles/r01_gemm_ablation.txt): MFMA alone 2.0k cycles per K tile, MFMA + LDS reads + barrier
// 2.6k, fills alone 2.3k, everything together 3.4k -- the three activities serialise because all eight
// waves do the same thing at the same time.  Here:
//   * waves are 2 (M) x 4 (N), each owning 128 x 64 of C; the SIMD that hosts wave w also hosts wave w+4,
//     i.e. one wave of each M group.  Group 1 runs ONE barrier behind group 0, so while one wave of a
//     SIMD issues its 8 MFMAs (s_setprio 1) the other one is in its load section (LDS-DMA issue,
//     ds_read_b128, s_waitcnt) -- matrix pipe and LDS/TA work overlap instead of alternating;
//   * a K tile is consumed in 4 phases, one C quadrant (64 rows x 32 cols of the wave tile, K = BK) each:
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0); A-half h = the rows of quadrant row h of both M groups, B-half
//     h likewise for the four N groups.  A half-tile (128 rows x 128 B = 16 KB = 2 LDS-DMA ops per
//     thread) is refilled for K tile t+2 as soon as its last reader is two barriers behind, so fills
//     run up to 1.75 K tiles ahead and are issued one half-tile per phase;
//   * the memory waits in the loop are counted (`s_waitcnt vmcnt(8)` / `vmcnt(4)`, never 0 in steady state):
//     vmcnt retires in order, so "all but the N newest LDS-DMA ops" == "the half-tiles needed next have landed".
// Hazard rules used (two wave groups one barrier apart, cf. cdna_hip_programming.md 8-phase notes):
//   RAW  a half-tile waited for (vmcnt) in phase p may be read from phase p+1 on;
//   WAR  a slot whose last ds_read was issued in phase p may be refilled from phase p+2 on (the read itself
//        is only retired after that phase's first barrier).
#pragma once
#include "gemm.h"

namespace plipmi {

template <typename T, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_nt_8phase_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256;
  constexpr int ELEMS16 = 16 / sizeof(T);
  constexpr int BK = 8 * ELEMS16;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int MI = 4, NI = 2;  // wave tile 128 x 64

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile assignment (XCD strips + column groups, as gemm.h) -------------------------------
  const int nbn = p.N / BN;
  const int nbm = (p.M + BM - 1) / BM;
  const int nblk = nbm * nbn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, xi = bid >> 3, xq = nblk >> 3, xr = nblk & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int gw = p.gw > 0 ? p.gw : nbn, tpg = nbm * gw;
  const int cgrp = lid / tpg, crem = lid - cgrp * tpg;
  const int m0 = (crem / gw) * BM, n0 = (cgrp * gw + crem % gw) * BN;

  // ---- half-tile fills: half h of A = rows {g*128 + h*64 + [0,64)}, g = 0,1 (one LDS-DMA op each);
  //      half h of B = rows {q*64 + h*32 + [0,32)}, q = 0..3 (op g covers q = 2g, 2g+1) ----------------
  const int l8 = lane >> 3;
  const char* a_src[2][2];
  const char* w_src[2][2];
  unsigned a_dst[2][2], w_dst[2][2];  // wave-uniform LDS byte offsets inside a stage
  {
    const int arow = wave * 8 + l8;                       // 0..63 inside the 64-row piece
    const int asw = (arow >> 1) & 7;
    const int brow = (wave & 3) * 8 + l8;                 // 0..31 inside the 32-row piece
    const int bsw = (brow >> 1) & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int r = m0 + g * 128 + h * 64 + arow;
        r = r < p.M ? r : p.M - 1;
        a_src[h][g] = reinterpret_cast<const char*>(p.A) + (size_t)r * p.lda * sizeof(T) + (((lane & 7) ^ asw) << 4);
        a_dst[h][g] = (g * 128 + h * 64 + wave * 8) * 128;
        const int q = 2 * g + (wave >> 2);
        const int rn = n0 + q * 64 + h * 32 + brow;
        w_src[h][g] = reinterpret_cast<const char*>(p.W) + (size_t)rn * p.ldw * sizeof(T) + (((lane & 7) ^ bsw) << 4);
        w_dst[h][g] = A_BYTES + (q * 64 + h * 32 + (wave & 3) * 8) * 128;
      }
  }
  const unsigned lds0 =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  auto fill_a = [&](int h, int buf) {
#pragma unroll
    for (int g = 0; g < 2; ++g) { glds16(a_src[h][g], lds0 + buf * STAGE + a_dst[h][g]); a_src[h][g] += 128; }
  };
  auto fill_w = [&](int h, int buf) {
#pragma unroll
    for (int g = 0; g < 2; ++g) { glds16(w_src[h][g], lds0 + buf * STAGE + w_dst[h][g]); w_src[h][g] += 128; }
  };

  // ---- fragments --------------------------------------------------------------------------
  const int lrow = lane & 31, lgrp = lane >> 5;
  const int lsw = (lrow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lrow * 128 + (((ks * 2 + lgrp) ^ lsw) << 4);
  const int a_tile = wm * 128 * 128;
  const int w_tile = A_BYTES + wn * 64 * 128;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  u32x4 af[2][4];      // current A quadrant: [row tile of the quadrant][K step]
  u32x4 bf0[4], bf1[4];  // B quadrants 0 and 1

  auto read_a = [&](int buf, int qa) {
    const char* sb = smem + buf * STAGE + a_tile + qa * 64 * 128;
#pragma unroll
    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[ml][ks] = *reinterpret_cast<const u32x4*>(sb + ml * 32 * 128 + foff[ks]);
  };
  auto read_b = [&](int buf, int qb, u32x4 (&bf)[4]) {
    const char* sb = smem + buf * STAGE + w_tile + qb * 32 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const u32x4*>(sb + foff[ks]);
  };
  // end of a load section: the ds_reads are only ISSUED here; they are waited for after the barrier, so their
  // latency overlaps the barrier wait / the tail of the partner group's MFMA section
  auto bar_load = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_q = [&](int qa, int qb, const u32x4 (&bf)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ml = 0; ml < 2; ++ml) mma16<T>(acc[qa * 2 + ml][qb], bf[ks], af[ml][ks]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int KT = p.K / BK;
  unsigned long long* trace = p.trace ? p.trace + (size_t)bid * 8 : nullptr;
  if (trace && tid == 0) {
    trace[0] = __builtin_amdgcn_s_memtime();
    trace[4] = lid;
    trace[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
    trace[6] = KT;
  }

  // ---- prologue: K tile 0 complete, A0/B0/B1 of K tile 1 in flight (A1 of tile 1 goes out in phase 1) ----
  fill_a(0, 0); fill_w(0, 0); fill_w(1, 0); fill_a(1, 0);
  if (KT > 1) {
    fill_a(0, 1); fill_w(0, 1); fill_w(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    wait_vm0();
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();            // RAW rule: reads start two barriers after the wait
  if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 now runs one barrier behind group 0
  if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();

  // Slot reuse (refill >= 2 phases after the slot's last ds_read was issued) and the waits (>= 1 phase before
  // the first read of the data):
  //   phase 1 reads A0,B0(t)   issues A1(t+1)            [A1 slot of the other stage: last read phase 3 of t-1]
  //   phase 2 reads B1(t)      waits A1(t)     vmcnt(8)  [younger: A0,B0,B1,A1 of t+1]
  //   phase 3 reads A1(t)      issues A0(t+2)            [A0 slot: last read phase 1]
  //   phase 4 reads -          waits A0,B0,B1(t+1) vmcnt(4) [younger: A1(t+1), A0(t+2)]; issues B0,B1(t+2)
  for (int t = 0; t < KT; ++t) {
    const int buf = t & 1;
    const bool n1 = t + 1 < KT, n2 = t + 2 < KT;
    // phase 1: quadrant (A0, B0)
    if (n1) fill_a(1, buf ^ 1);
    read_a(buf, 0);
    read_b(buf, 0, bf0);
    bar_load();
    mma_q(0, 0, bf0);
    // phase 2: (A0, B1)
    if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else wait_vm0();
    read_b(buf, 1, bf1);
    bar_load();
    mma_q(0, 1, bf1);
    // phase 3: (A1, B1)
    if (n2) fill_a(0, buf);
    read_a(buf, 1);
    bar_load();
    mma_q(1, 1, bf1);
    // phase 4: (A1, B0)
    if (n1) {
      if (n2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if (n2) { fill_w(0, buf); fill_w(1, buf); }
    bar_load();
    mma_q(1, 0, bf0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the groups
  if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue (as gemm.h): per-wave 32 x 64 fp32 slabs, row-contiguous loads / stores ----------
  constexpr int SLAB_PITCH = 64 * 4 + 16;
  constexpr int SLAB_BYTES = 32 * SLAB_PITCH;
  static_assert(8 * SLAB_BYTES <= 2 * STAGE, "epilogue slabs must fit in the staging buffers");
  wait_vm0();
  __syncthreads();
  char* slab = smem + wave * SLAB_BYTES;
  const int rd_row = lane >> 4, rd_col = (lane & 15) * 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float4 add[8];
    const int n = n0 + wn * 64 + rd_col;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + wm * 128 + i * 32 + it * 4 + rd_row;
      add[it] = EpilogueOp<T, EPI>::load(p, m < p.M ? m : p.M - 1, n);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[i][jj][4 * q + 0], acc[i][jj][4 * q + 1], acc[i][jj][4 * q + 2], acc[i][jj][4 * q + 3]};
        *reinterpret_cast<f32x4*>(slab + lrow * SLAB_PITCH + (jj * 32 + 8 * q + 4 * lgrp) * 4) = v;
      }
    __builtin_amdgcn_wave_barrier();
    f32x4 v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
      v[it] = *reinterpret_cast<const f32x4*>(slab + (it * 4 + rd_row) * SLAB_PITCH + rd_col * 4);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + wm * 128 + i * 32 + it * 4 + rd_row;
      if (m < p.M) EpilogueOp<T, EPI>::store(p, m, n, v[it][0], v[it][1], v[it][2], v[it][3], add[it]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (trace) {
    __builtin_amdgcn_s_waitcnt(0);
    if (tid == 0) trace[3] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace plipmi
