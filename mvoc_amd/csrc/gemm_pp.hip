// Persistent "ping-pong" implicit GEMM for gfx950: the large-M launches of the network (every linear / 3x3 conv /
// temporal conv with >= 64 output tiles of 256 pixels) run through this kernel.
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )      fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16)
//
// Same math, operand conventions, gathers (PLAIN / CONV3X3 / TEMPORAL3, two sources, folded upsample) and the same fp16
// rounding points in the epilogue as gemm.hip; what differs is the structure around the MFMAs:
//
// * ONE block of 8 waves per CU, persistent: the block walks a list of work units (m-tile, n-tile, K-slice) -- tile =
//   256 pixels x 256 (or 320) output channels.  The operand ring keeps running across unit boundaries, so the first K
//   steps of unit u+1 are already in LDS when unit u's epilogue ends (the K = 320 / 640 projections of the two finest
//   levels have only 10-20 K steps per tile: per-block prologue / epilogue used to cost them half their time).
// * K step 32 per ring stage (64-byte rows, XOR-swizzled as in gemm.hip), NS = 4 (3) stages, filled by LDS-DMA
//   (global_load_lds_dwordx4) that stays in flight ACROSS barriers: counted s_waitcnt vmcnt(N) + raw s_barrier, never a
//   __syncthreads() in the loop.
// * The two wave groups of the block (waves 0-3 = output channels 0..BN/2, waves 4-7 = the other half; one wave of each
//   group per SIMD) run the same program ONE BARRIER APART:
//       group A:      READ(s)  | MFMA(s)  | READ(s+1) | MFMA(s+1) | ...
//       group B:   -  |  READ(s)  | MFMA(s)  | READ(s+1) | MFMA(s+1) ...
//   READ(s) = the wave's 12-14 ds_read_b128 fragment reads of stage s + its share of the LDS-DMA issue for stage s+NS-1 +
//   pointer bookkeeping; MFMA(s) = 16-20 back-to-back MFMAs on registers only.  In every interval one wave of each SIMD
//   owns the matrix pipe while its partner owns the LDS / address / VMEM-issue side, by construction rather than by
//   scheduler luck (two waves running the same one-barrier loop tend to fall into lockstep: MI355X_MICROARCH.md, "Two
//   waves per SIMD", item 9).
//   Hazards: a stage is read one barrier after every wave's counted vmcnt retired its pieces (RAW); a ring slot is
//   refilled only after both groups' reads of it were waited for (lgkmcnt(0) before the barrier that ends a READ) (WAR).
// * Epilogue per wave through a private 4 KB LDS tile (XOR-swizzled, no padding): 128-byte-contiguous stores / residual
//   loads per pixel row.  No block barrier inside: while group A stores, group B still multiplies, and vice versa.
#include "gemm_args.h"

namespace {

constexpr int KS = 32;      // K per ring stage
constexpr int ROWB = 64;    // bytes per staged row
__device__ const uint4 g_zero16_pp = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// In-kernel stamps (diagnostic instantiation only): s_memtime + its own lgkmcnt(0) as ONE statement, fenced for the scheduler
#define PP_STAMP(t)                                                                            \
  do {                                                                                         \
    if constexpr (LAB) {                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");                 \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }                                                                                          \
  } while (0)

// SPLIT = how many of the wave's two activation pieces per stage are issued from inside its MFMA interval (between the 8th
// and 9th MFMA) instead of its READ interval: a piece costs the issuing wave ~100 cycles, and four in a row made READ twice
// as long as MFMA (in-kernel stamps, tools/pp_lab.py)
// BUF: the LDS-DMA as MUBUF (buffer_load_dwordx4 ... offen lds: wave-uniform 128-bit resource + one 32-bit byte offset per lane)
// instead of FLAT-global (64-bit address per lane).  Padding rows / taps outside the image / rows past M or N use an offset
// beyond the resource's range: the hardware range check returns zeros, no zero page and no per-lane step are needed.
template <int TN, int NS, bool LAB = false, int SPLIT = 0, bool BUF = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs p, const int n_units) {
  constexpr int WM = 4, TM = 2;
  constexpr int BN = 2 * TN * 32, BM = WM * TM * 32;
  constexpr int PCW = BN / 16;                 // 1 KB pieces (16 rows x 64 B) of the weight tile per stage
  constexpr int PWMAX = (PCW + 7) / 8;         // per wave (piece j = wave + 8 i)
  constexpr int STAGE = (BN + BM) * ROWB;
  constexpr int RING = NS * STAGE;
  constexpr int EPI = 4096;                    // per-wave epilogue tile: 32 pixels x 64 channels fp16
  static_assert(RING + 8 * EPI <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char smem[RING + 8 * EPI];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wm = wave & 3;     // group = wn
  const int r = lane & 31, h = lane >> 5;
  const half_t* zsrc = reinterpret_cast<const half_t*>(&g_zero16_pp);
  // LDS-DMA pieces per wave and stage (wave-uniform, compile-time per group for BN = 256)
  const int npw = (PCW - wave + 7) / 8;        // valid weight pieces of this wave (2; 3 or 2 for BN = 320)
  const int S = p.k_per_split / KS;            // stages per unit

  // ---- unit list of this block: logical units G apart, blocks of one XCD on neighbouring units -----------------
  const int G = (int)gridDim.x;
  const int bperm = (G % 8 == 0) ? (int)(blockIdx.x & 7u) * (G / 8) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int my_units = (n_units - bperm + G - 1) / G;  // >= 1 (grid <= n_units)
  const int T = my_units * S;                          // ring stages this block consumes

  // ---- producer state ---------------------------------------------------------------------------------------------
  const int lrow = lane >> 2;                     // row inside a piece
  const int cch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;  // source chunk (halfs) for LDS position lane&3: swizzle (row>>2)&3
  const half_t* wptr[PWMAX];
  int wstep[PWMAX];
  const half_t* aptr[2];
  int astep[2];
  unsigned woff[PWMAX], aoff[2];  // BUF: byte offsets into the weight / activation resources
  constexpr unsigned OOB = 0x80000000u;  // >= num_records: the load returns zeros
  bool a_second = false;
  __amdgpu_buffer_rsrc_t rs_w, rs_a, rs_a2;
  if constexpr (BUF) {
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)OOB, 0x00020000);
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)OOB, 0x00020000);
    rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, (int)OOB, 0x00020000);
  }
  int rowoff[2];
  unsigned vmask[2];
  int ry0[2], rx0[2], rimg[2];
  int p_unit = 0, p_kt = 0, p_tap = 0, p_ch0 = 0;
  bool regather = true;

  auto decode = [&](int k, int& m0, int& n0, int& slice) {
    const int u = bperm + k * G;
    slice = u % p.split_k;
    const int t = u / p.split_k;
    n0 = (t % p.n_tiles) * BN;
    m0 = (t / p.n_tiles) * BM;
  };

  auto setup_unit = [&](int k) {
    int m0, n0, slice;
    decode(k, m0, n0, slice);
    const int kbeg = slice * p.k_per_split;
#pragma unroll
    for (int i = 0; i < PWMAX; ++i) {
      const int row = (wave + 8 * i) * 16 + lrow;
      const bool ok = row < BN && n0 + row < p.N;
      if constexpr (BUF) {
        woff[i] = ok ? (unsigned)(((size_t)(n0 + row) * p.K + cch + kbeg) * 2) : OOB;
      } else {
        wptr[i] = ok ? p.w + (size_t)(n0 + row) * p.K + cch + kbeg : zsrc;
        wstep[i] = ok ? KS : 0;
        if (LAB && (p.lab & 2)) { wptr[i] = zsrc; wstep[i] = 0; }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 8 * i) * 16 + lrow;
      const int m = m0 + row;
      const bool live = m < p.M;
      const int mm = live ? m : 0;
      ry0[i] = rx0[i] = rimg[i] = 0;
      rowoff[i] = mm;
      vmask[i] = live ? 1u : 0u;
      if (p.a_mode == MVOC_A_CONV3X3) {
        const int hwout = p.hout * p.wout;
        const int img = mm / hwout;
        const int rem = mm - img * hwout;
        const int oy = rem / p.wout;
        const int y0 = oy * p.stride - p.pad, x0 = (rem - oy * p.wout) * p.stride - p.pad;
        rimg[i] = img; ry0[i] = y0; rx0[i] = x0;
        rowoff[i] = (img * p.hsrc + y0) * p.wsrc + x0;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y0 + t / 3, ix = x0 + t % 3;
          if (live && iy >= 0 && ix >= 0 && iy < p.hup && ix < p.wup) mk |= 1u << t;
        }
        vmask[i] = mk;
      } else if (p.a_mode == MVOC_A_TEMPORAL3) {
        const int f = (mm / p.hw) % p.frames;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (live && f + t - 1 >= 0 && f + t - 1 < p.frames) mk |= 1u << t;
        vmask[i] = mk;
      }
    }
    p_tap = kbeg / p.cin;
    p_ch0 = kbeg - p_tap * p.cin;
    regather = true;
  };

  bool need_setup = false;
  auto issue_w = [&](int slot) {  // weight pieces of the next stage + (at a unit boundary) the new unit's row bookkeeping
    if (need_setup) { setup_unit(p_unit); need_setup = false; }
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < PWMAX; ++i) {
      if constexpr (BUF) {
        if (i < npw)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(base + (wave + 8 * i) * 1024), 16,
                                                   (int)woff[i], 0, 0, 0);
        woff[i] += KS * 2;  // (an out-of-range offset stays out of range)
      } else {
        if (i < npw)  // wave-uniform
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wptr[i],
                                           (__attribute__((address_space(3))) void*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
        wptr[i] += wstep[i];
      }
    }
    if (regather) {  // wave-uniform: first stage of a unit, a new tap, or the switch to the second source
      const bool second = p_ch0 >= p.c1;
      const half_t* sbase = second ? p.a2 : p.a;
      const int ld = second ? p.lda2 : p.lda;
      const int cbase = second ? p_ch0 - p.c1 : p_ch0;
      const int ky = p_tap / 3, kx = p_tap - ky * 3;
      int tap_rows = 0;
      if (p.a_mode == MVOC_A_CONV3X3) tap_rows = ky * p.wsrc + kx;
      else if (p.a_mode == MVOC_A_TEMPORAL3) tap_rows = (p_tap - 1) * p.hw;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = (vmask[i] >> p_tap) & 1u;
        long srow = rowoff[i] + tap_rows;
        if (p.upsample) {  // nearest-upsampled source: the row is not affine in the tap
          const int iy = min((int)floorf((ry0[i] + ky) * p.ups_sh), p.hsrc - 1);
          const int ix = min((int)floorf((rx0[i] + kx) * p.ups_sw), p.wsrc - 1);
          srow = ((long)rimg[i] * p.hsrc + iy) * p.wsrc + ix;
        }
        if constexpr (BUF) {
          aoff[i] = ok ? (unsigned)((srow * ld + (cbase + cch)) * 2) : OOB;
          a_second = second;
        } else {
          aptr[i] = ok ? sbase + srow * ld + (cbase + cch) : zsrc;
          astep[i] = ok ? KS : 0;
          if (LAB && (p.lab & 2)) { aptr[i] = zsrc; astep[i] = 0; }
        }
      }
    }
  };
  auto issue_a = [&](int slot, int i) {
    if constexpr (BUF) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_second ? rs_a2 : rs_a,
                                               (__attribute__((address_space(3))) void*)(smem + slot * STAGE + BN * ROWB + (wave + 8 * i) * 1024),
                                               16, (int)aoff[i], 0, 0, 0);
      aoff[i] += KS * 2;
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)aptr[i],
                                       (__attribute__((address_space(3))) void*)(smem + slot * STAGE + BN * ROWB + (wave + 8 * i) * 1024), 16, 0, 0);
      aptr[i] += astep[i];
    }
  };
  auto end_stage = [&]() {  // scalar bookkeeping only
    p_ch0 += KS;
    regather = p_ch0 == p.c1;
    if (p_ch0 >= p.cin) { p_ch0 = 0; ++p_tap; regather = true; }
    if (++p_kt == S) {
      p_kt = 0;
      if (++p_unit < my_units) need_setup = true;
    }
  };
  auto issue = [&](int slot) {
    issue_w(slot);
    issue_a(slot, 0);
    issue_a(slot, 1);
    end_stage();
  };

  // counted waits: this wave issues P = npw + 2 pieces per stage
  auto wait_stages = [&](int stages_in_flight) {  // all but the newest `stages_in_flight` stages of this wave have landed
    if constexpr (TN == 4) {
      if (stages_in_flight >= 2) wait_vm<8>(); else if (stages_in_flight == 1) wait_vm<4>(); else wait_vm<0>();
    } else {
      if (npw == 3) {
        if (stages_in_flight >= 2) wait_vm<10>(); else if (stages_in_flight == 1) wait_vm<5>(); else wait_vm<0>();
      } else {
        if (stages_in_flight >= 2) wait_vm<8>(); else if (stages_in_flight == 1) wait_vm<4>(); else wait_vm<0>();
      }
    }
  };

  // ---- consumer state ------------------------------------------------------------------------------------------------
  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int swz = (r >> 2) & 3;
  char* epi = smem + RING + wave * EPI;

  // ---- start-up: NS-1 stages in flight, stage 0 landed and visible; group B falls one barrier behind ------------------
  setup_unit(0);
#pragma unroll 1
  for (int g = 0; g < NS - 1 && g < T; ++g) issue(g);
  wait_stages(T >= NS - 1 ? NS - 2 : 0);
  __builtin_amdgcn_s_barrier();
  if (wn == 1) __builtin_amdgcn_s_barrier();

  int c_unit = 0, c_kt = 0, slot = 0;
  bool skip_wait = false;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
  unsigned long long sum_rd = 0, sum_is = 0, sum_vw = 0, sum_b1 = 0, sum_mf = 0, sum_b2 = 0, sum_ep = 0, sum_all = 0;
  const bool no_dma = LAB && (p.lab & 1);
  (void)no_dma;
  PP_STAMP(ts6);
  const unsigned long long t_begin = ts6;
#pragma unroll 1
  for (int g = 0; g < T; ++g) {
    // ================= READ(g) =================
    PP_STAMP(ts0);
    half8_t wf[2][TN], af[2][TM];
    {
      const char* wl = smem + slot * STAGE + (wn * TN * 32 + r) * ROWB;
      const char* al = smem + slot * STAGE + BN * ROWB + (wm * TM * 32 + r) * ROWB;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int off = ((2 * s + h) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[s][i] = *reinterpret_cast<const half8_t*>(wl + i * 32 * ROWB + off);
#pragma unroll
        for (int j = 0; j < TM; ++j) af[s][j] = *reinterpret_cast<const half8_t*>(al + j * 32 * ROWB + off);
      }
    }
    PP_STAMP(ts1);  // (diagnostic: the fragment reads are drained here)
    if (g + NS - 1 < T) {
      int fs = slot + NS - 1;
      if (fs >= NS) fs -= NS;
      // the slot stage g-1 lived in: both groups' reads of it were waited for before their last barrier
      if (!no_dma) {
        issue_w(fs);
#pragma unroll
        for (int i = 0; i < 2 - SPLIT; ++i) issue_a(fs, i);
        if constexpr (SPLIT == 0) end_stage();
      }
      PP_STAMP(ts2);
      if (!skip_wait) {  // stage g+1 of this wave's pieces has landed; SPLIT pieces of stage g+NS-1 are not issued yet
        if constexpr (SPLIT == 0) wait_stages(NS - 2);
        else { static_assert(SPLIT == 0 || (TN == 4 && NS == 4), "split issue: 256-wide tile"); wait_vm<8 - SPLIT>(); }
      }
    } else {
      PP_STAMP(ts2);
      if (!skip_wait) wait_stages(0);
    }
    skip_wait = false;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(ts3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(ts4);
    // ================= MFMA(g) =================
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s][i], af[s][j], acc[i][j], 0, 0, 0);
      if constexpr (SPLIT > 0) {
        if (s == 0 && g + NS - 1 < T && !no_dma) {
          int fs = slot + NS - 1;
          if (fs >= NS) fs -= NS;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 2 - SPLIT; i < 2; ++i) issue_a(fs, i);
          end_stage();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    PP_STAMP(ts5);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(ts6);
    if constexpr (LAB) {
      sum_rd += ts1 - ts0; sum_is += ts2 - ts1; sum_vw += ts3 - ts2; sum_b1 += ts4 - ts3; sum_mf += ts5 - ts4; sum_b2 += ts6 - ts5;
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
    if (++c_kt < S) continue;
    // ================= unit done: epilogue (no block barrier inside) =================
    c_kt = 0;
    // the post-epilogue READ skips its wait (the stores issued below would sit in front of it): retire stage g+2 here
    if (g + NS - 1 < T) wait_stages(NS - 3 > 0 ? NS - 3 : 0); else wait_stages(0);
    skip_wait = true;
    int m0, n0, slice;
    decode(c_unit, m0, n0, slice);
    ++c_unit;
    if (p.split_k > 1) {  // raw fp32 partials; bias / activation / residual happen in the reduce pass
      float* slab = p.ws + (size_t)slice * p.M * p.N;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 32 + r;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;
            if (m < p.M && n < p.N) {
              f32x4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
              *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = v;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = 0.f;
          }
      }
      continue;
    }
    const bool geglu = p.act == MVOC_ACT_GEGLU;
    const bool use_bias = p.bias && !p.ln_s;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int mrow = m0 + (wm * TM + j) * 32;
      const int m_own = mrow + r;
      const int m_safe = m_own < p.M ? m_own : p.M - 1;
      const half_t* ra = p.rowadd ? p.rowadd + (size_t)(m_safe / p.rowadd_div) * p.ld_rowadd : nullptr;
      float ln_mu = 0.f, ln_rs = 1.f;
      if (p.ln_s) {
        ln_mu = p.ln_stats[2 * (size_t)m_safe];
        ln_rs = p.ln_stats[2 * (size_t)m_safe + 1];
        // consume the pair here: a load still pending on some path at the loop's back edge makes hipcc drain vmcnt(0)
        // in front of the K loop's fragment reads (its destination registers are reused there)
        asm volatile("" ::"v"(ln_mu), "v"(ln_rs));
      }
      // a round parks 32 pixels x 64 output channels (fp16) in the wave's LDS tile (16-byte chunk position XORed with the
      // pixel row: conflict-free enough without padding), reads them back as 128-byte rows, adds the residual, stores
      auto drain = [&](int nbase, int nvalid) {  // nvalid: output channels of this round that exist (32 or 64)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = lane + it * 64;
          const int px = idx >> 3, c = idx & 7;
          const int m = mrow + px, n = nbase + c * 8;
          const bool on = m < p.M && n < p.n_store && c * 8 < nvalid;
          half8_t v = *reinterpret_cast<const half8_t*>(epi + px * 128 + ((c ^ (px & 7)) * 16));
          if (on) {
            if (p.resid) {
              const half8_t r8 = *reinterpret_cast<const half8_t*>(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r8[e]);
            }
            *reinterpret_cast<half8_t*>(p.out + (size_t)m * p.ldo + n) = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next round overwrites the tile
      };
      if (geglu) {
        if constexpr (TN % 2 == 0) {
#pragma unroll
          for (int i0 = 0; i0 < TN; i0 += 4) {  // 4 MFMA tiles = 2 (value, gate) pairs = 64 output channels
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = i0 + 2 * u;
              if (i + 1 >= TN) continue;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                half4_t o = {0, 0, 0, 0};
                const int nh = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;  // packed row of the value half
                if (nh < p.N) {
                  half4_t bh = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
                  if (use_bias) {
                    bh = *reinterpret_cast<const half4_t*>(p.bias + nh);
                    bg = *reinterpret_cast<const half4_t*>(p.bias + nh + 32);
                  }
                  f32x4 sh = {0.f, 0.f, 0.f, 0.f}, sg = sh, ch = sh, cg = sh;
                  if (p.ln_s) {
                    sh = *reinterpret_cast<const f32x4*>(p.ln_s + nh);
                    sg = *reinterpret_cast<const f32x4*>(p.ln_s + nh + 32);
                    ch = *reinterpret_cast<const f32x4*>(p.ln_c + nh);
                    cg = *reinterpret_cast<const f32x4*>(p.ln_c + nh + 32);
                  }
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float hv = r16(p.ln_s ? ln_rs * (acc[i][j][q * 4 + e] - ln_mu * sh[e]) + ch[e] : acc[i][j][q * 4 + e] + (float)bh[e]);
                    const float gv = r16(p.ln_s ? ln_rs * (acc[i + 1][j][q * 4 + e] - ln_mu * sg[e]) + cg[e]
                                                : acc[i + 1][j][q * 4 + e] + (float)bg[e]);
                    o[e] = (half_t)(hv * r16(gelu_fast_f(gv)));
                  }
                }
                *reinterpret_cast<half4_t*>(epi + r * 128 + (((u * 4 + q) ^ (r & 7)) * 16) + h * 8) = o;
              }
            }
            drain((n0 + (wn * TN + i0) * 32) / 2, i0 + 3 < TN ? 64 : 32);
          }
        }
      } else {
#pragma unroll
        for (int i0 = 0; i0 < TN; i0 += 2) {  // 2 MFMA tiles = 64 output channels
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = i0 + u;
            if (i >= TN) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              half4_t o = {0, 0, 0, 0};
              const int n = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;
              if (n < p.n_store) {
                half4_t b4 = {0, 0, 0, 0};
                if (use_bias) b4 = *reinterpret_cast<const half4_t*>(p.bias + n);
                float v[4];
                if (p.ln_s) {
                  const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
                  const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.ln_c + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = r16(ln_rs * (acc[i][j][q * 4 + e] - ln_mu * s4[e]) + c4[e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = r16(acc[i][j][q * 4 + e] + (float)b4[e]);
                }
                if (ra) {
                  const half4_t t4 = *reinterpret_cast<const half4_t*>(ra + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)t4[e]);
                }
                if (p.act == MVOC_ACT_SILU) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = r16(silu_f(v[e]));
                } else if (p.act == MVOC_ACT_GELU) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = r16(gelu_fast_f(v[e]));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
              }
              *reinterpret_cast<half4_t*>(epi + r * 128 + (((u * 4 + q) ^ (r & 7)) * 16) + h * 8) = o;
            }
          }
          drain(n0 + (wn * TN + i0) * 32, i0 + 1 < TN ? 64 : 32);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  if (wn == 0) __builtin_amdgcn_s_barrier();  // group A's balancing barrier (B ran one extra at the start)
  if constexpr (LAB) {
    PP_STAMP(ts0);
    sum_all = ts0 - t_begin;
    sum_ep = sum_all - (sum_rd + sum_is + sum_vw + sum_b1 + sum_mf + sum_b2);
    if (p.stamps && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
      unsigned long long* o = p.stamps + (wave >> 2) * 8;
      o[0] = sum_rd; o[1] = sum_is; o[2] = sum_vw; o[3] = sum_b1; o[4] = sum_mf; o[5] = sum_b2; o[6] = sum_ep; o[7] = sum_all;
    }
  }
}

template <int TN, int NS, bool LAB = false, int SPLIT = 0, bool BUF = false>
int launch_pp(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  constexpr int BN = 2 * TN * 32, BM = 256;
  a.n_tiles = (a.N + BN - 1) / BN;
  a.m_tiles = (a.M + BM - 1) / BM;
  const long units = (long)a.n_tiles * a.m_tiles * a.split_k;
  if (units <= 0 || units > 0x3fffffffL) {
    mvoc_set_error("gemm_pp: %ld work units", units);
    return -2;
  }
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    ncu = v;
  }
  const int grid = (int)(units < ncu ? units : ncu);
  hipLaunchKernelGGL((gemm_pp_kernel<TN, NS, LAB, SPLIT, BUF>), dim3((unsigned)grid), dim3(512), 0, s, a, (int)units);
  return mvoc_check_launch("gemm_pp_kernel");
}

}  // namespace

int mvoc_launch_gemm_pp(const GemmArgs& a, int bn, hipStream_t s) {
#ifdef MVOC_PP_LAB
  if (bn == 256 && (a.lab || a.stamps)) return launch_pp<4, 4, true>(a, s);
#endif
  if (bn == 256) return launch_pp<4, 4>(a, s);
  if (bn == 2561) return launch_pp<4, 4, false, 1>(a, s);
  if (bn == 2562) return launch_pp<4, 4, false, 2>(a, s);
  if (bn == 2563) return launch_pp<4, 4, false, 0, true>(a, s);
  if (bn == 320) return launch_pp<5, 3>(a, s);
  mvoc_set_error("gemm_pp: unsupported tile width %d", bn);
  return -1;
}
