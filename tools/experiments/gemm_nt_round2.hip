// Round-2 gemm_nt experiments that LOST to the production kernels (gemm_nt.hip) and are not part of libclipa_hip.so.
// Kept as measured baselines; results in profiles/r02_gemm_epilogue_experiments.md.  They compile against
// gemm_common.h and the helpers of gemm_nt.hip (bperm, store_direct, epilogue_direct, sink_acc): paste them back
// above `g_nt_once` in gemm_nt.hip to re-run tools/gemm_round2.py (git history of round 2 has the wired-up version,
// including the KEEP mode of the direct epilogue that gemm_nte calls as epilogue_direct<EPI, HAS_C2, true>).
//   gemm_nte  direct epilogue + half of the packed tile stored during the next tile's main loop   (variant 15)
//   gemm_nt6  256x128x32 tile, 4 waves, two persistent workgroups per CU                            (variant 13)
//   gemm_nt7  256x256 tile, K step 32, five-slot ring with four steps in flight, counted vmcnt      (variant 14)

//   gemm_ntd  direct (MFMA-fragment-layout) epilogue on the production tile                         (variant 12)

// ------------------------------------------------------------------------------------------------
// DIRECT epilogue (gemm_ntd / gemm_nt6).  LDS row r = 16*bj + i of a wave's 64 B-image rows (MFMA row i of its
// n block bj) is filled, at DMA time, with tile feature
//     bperm(r) = 32*(bj>>1) + 8*(i>>2) + 4*(bj&1) + (i&3).
// An MFMA leaves rows 4g..4g+3 of block bj in lane (c = lane & 15, g = lane >> 4), so for s = 0, 1 that lane owns
// the 8 CONSECUTIVE features 32*s + 8*g + 0..7 (blocks 2s and 2s+1) of token row 16*ai + c:
//     acc[bj][ai][r]  ==  C[mw + 16*ai + c][nw + 32*(bj>>1) + 8*g + 4*(bj&1) + r]
// One 16-byte buffer store per (ai, s): 16 rows x 64 contiguous bytes per instruction, no LDS transpose, no
// barrier.  Row bounds ride the buffer descriptor (the row offset sits in the bounds-checked VGPR offset: rows >= M
// are dropped / read as zero), the column bound is a lane
// predicate.  Rounding points are those of the LDS-window epilogue (and of the reference in bf16 mode: a Linear's
// output is rounded to bf16 before the activation / the residual add sees it).
__device__ __forceinline__ int bperm(int r64) {
  const int bj = r64 >> 4, i = r64 & 15;
  return 32 * (bj >> 1) + 8 * (i >> 2) + 4 * (bj & 1) + (i & 3);
}

template <int EPI, bool HAS_C2, int ACT>
__device__ __forceinline__ void store_direct_act(const NTArgs& p, const f32x4v (&acc)[4][8], int mw, int nw, int lane) {
  constexpr bool NEED_AUX = (EPI == CLIPA_EPI_ADD || EPI == CLIPA_EPI_DACT);
  const int rows = min(128, p.M - mw);
  if (rows <= 0) return;                                  // wave-uniform
  const int c = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(p.C + (size_t)mw * p.ldc * 2, (unsigned)((long)rows * p.ldc * 2));
  __amdgpu_buffer_rsrc_t rsC2 = rsC, rsX = rsC;
  if constexpr (HAS_C2) rsC2 = make_rsrc(p.C2 + (size_t)mw * p.ldc * 2, (unsigned)((long)rows * p.ldc * 2));
  if constexpr (NEED_AUX) rsX = make_rsrc(p.aux + (size_t)mw * p.ldaux * 2, (unsigned)((long)rows * p.ldaux * 2));
  const int stepC = (int)(16 * p.ldc * 2), stepX = (int)(16 * p.ldaux * 2);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int n = nw + 32 * s + 8 * g;
    const bool nok = n < p.N;
    const int voffC = (int)((c * p.ldc + n) * 2);
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok) {
      const float4 x = *(const float4*)(p.bias + n), y = *(const float4*)(p.bias + n + 4);
      bb[0] = x.x; bb[1] = x.y; bb[2] = x.z; bb[3] = x.w; bb[4] = y.x; bb[5] = y.y; bb[6] = y.z; bb[7] = y.w;
    }
    u32x4 av[8];
    if constexpr (NEED_AUX) {
      const int voffX = (int)((c * p.ldaux + n) * 2);
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) {
        av[ai] = u32x4{0, 0, 0, 0};
        if (nok) av[ai] = __builtin_amdgcn_raw_buffer_load_b128(rsX, voffX + ai * stepX, 0, 0);
      }
    }
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
      float f[8], a[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f[r] = acc[2 * s][ai][r] * p.alpha + bb[r];
        f[4 + r] = acc[2 * s + 1][ai][r] * p.alpha + bb[4 + r];
      }
      u32x4 v = pack8(f);
      if constexpr (EPI == CLIPA_EPI_ACT) {
        if constexpr (HAS_C2) {
          if (nok && !((p.abl & 1) && v[1] != 0x12345u)) __builtin_amdgcn_raw_buffer_store_b128(v, rsC2, voffC + ai * stepC, 0, 0);
        }
        unpack8(v, f);
        epi_apply<ACT>(CLIPA_EPI_ACT, f, a);
        v = pack8(f);
      } else if constexpr (EPI == CLIPA_EPI_ADD) {
        unpack8(v, f);
        unpack8(av[ai], a);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += a[i];
        v = pack8(f);
      } else if constexpr (EPI == CLIPA_EPI_DACT) {
        unpack8(v, f);
        unpack8(av[ai], a);
        epi_apply<ACT>(CLIPA_EPI_DACT, f, a);
        v = pack8(f);
      }
      if (nok && !((p.abl & 1) && v[0] != 0x12345u)) __builtin_amdgcn_raw_buffer_store_b128(v, rsC, voffC + ai * stepC, 0, 0);
    }
  }
}

template <int EPI, bool HAS_C2>
__device__ __forceinline__ void store_direct(const NTArgs& p, const f32x4v (&acc)[4][8], int mw, int nw, int lane) {
  if constexpr (EPI == CLIPA_EPI_ACT || EPI == CLIPA_EPI_DACT) {
    if (p.act == ACT_GELU_ERF) store_direct_act<EPI, HAS_C2, ACT_GELU_ERF>(p, acc, mw, nw, lane);
    else if (p.act == ACT_GELU_TANH) store_direct_act<EPI, HAS_C2, ACT_GELU_TANH>(p, acc, mw, nw, lane);
    else store_direct_act<EPI, HAS_C2, ACT_QUICK_GELU>(p, acc, mw, nw, lane);
  } else {
    store_direct_act<EPI, HAS_C2, ACT_GELU_ERF>(p, acc, mw, nw, lane);
  }
}

// keep the accumulators live without storing them (experiment flag 2: main loop alone)
__device__ __forceinline__ void sink_acc(const NTArgs& p, const f32x4v (&acc)[4][8], int tid) {
  float t = 0.f;
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) t += acc[bj][ai][0] + acc[bj][ai][1] + acc[bj][ai][2] + acc[bj][ai][3];
  if (t == 1.2345e-30f) ((float*)p.C)[tid] = t;
}

// ------------------------------------------------------------------------------------------------
// gemm_ntd: the 256x256x64 tile / 8 waves / 2-slot ring / 16x16x32 main loop of gemm_nt2<bf16, M16> with
//   * the operand DMA issued from inline asm (hidden from hipcc: no compiler vmcnt in front of LDS reads),
//   * the DIRECT epilogue above (B image rows permuted at DMA time),
//   * a ring hand-off that does not wait for the epilogue's stores: the first K tile of the next output tile is
//     waited for BEFORE the stores are issued (`vmcnt(0)` at the top of the epilogue: only DMA is outstanding
//     there), so the first main-loop iteration after an epilogue needs a barrier only; the stores have that
//     iteration (plus the epilogue's remainder) to drain before the next `vmcnt(0)`.
template <int EPI, bool HAS_C2>
__global__ __launch_bounds__(NTHREADS) void gemm_ntd_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;            // workgroups on this XCD
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  // DMA piece pc = j*8 + wave = image rows 8*pc .. 8*pc+7 (128 B each); lane -> row (lane>>3), physical chunk lane&7.
  // Image row 64*j + 8*wave + (lane>>3): its swizzle key (row>>1)&7 does not depend on j.
  const int r8l = lane >> 3;
  const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  const int kel = chunk * 8;
  const int i16 = 8 * (wave & 1) + r8l, bjw = wave >> 1;    // B image row 64*j + 16*bjw + i16 <- feature 64*j + bperm(.)
  const unsigned voffA = (unsigned)((wave * 8 + r8l) * p.lda * 2 + chunk * 16);
  const unsigned voffB = (unsigned)(bperm(16 * bjw + i16) * p.ldb * 2 + chunk * 16);
  const unsigned stepA = (unsigned)(64 * p.lda * 2), stepB = (unsigned)(64 * p.ldb * 2);
  const int nkt = (p.K + BK - 1) / BK;

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = (p.abl & 8) ? 1 : 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN, p.N - n) * p.ldb * 2)); };
  auto stage = [&](unsigned buf, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + kel >= p.K) ? 0x80000000u : 0u;   // K tail: push the lane past num_records -> zeros
    const unsigned dA = lds0 + buf * STAGE_BYTES + wave * 1024, dB = dA + IMG_BYTES;
    dma16_x4<8192>(rsA, dA, voffA | oob, (voffA + stepA) | oob, (voffA + 2 * stepA) | oob, (voffA + 3 * stepA) | oob, (unsigned)(k0 * 2));
    dma16_x4<8192>(rsB, dB, voffB | oob, (voffB + stepB) | oob, (voffB + 2 * stepB) | oob, (voffB + 3 * stepB) | oob, (unsigned)(k0 * 2));
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  u32x4 sa = srdA(m0), sb = srdB(n0);
  stage(0, sa, sb, 0);
  unsigned gk = 0;        // global K-tile counter: ring slot = gk & 1
  bool landed = false;    // K tile 0 of this output tile is already known to have landed (epilogue waited for it)
  const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);
    const u32x4 sa1 = srdA(m1), sb1 = srdB(n1);

    f32x4v acc[4][8];      // [n block of 16][m block of 16]
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      if (!(kt == 0 && landed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WG_BARRIER_LDS();
      if (kt + 1 < nkt) stage((gk + 1) & 1, sa, sb, (kt + 1) * BK);
      else if (has_next) stage((gk + 1) & 1, sa1, sb1, 0);
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      // 8 sub-steps per K tile: (kk, s) = 32-wide k-step kk, A blocks 2s and 2s+1 against the four B blocks of kk
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 64 + l15) * 128;
      bf16x8 ga[2][2], gb[2][4];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[0][bj] = *(const bf16x8*)(pb + bj * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int a = 0; a < 2; ++a) ga[0][a] = *(const bf16x8*)(pa + a * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = u >> 2, sbk = u & 3;
        if (u < 7) {
          const int k1 = (u + 1) >> 2, s1 = (u + 1) & 3;
#pragma unroll
          for (int a = 0; a < 2; ++a) ga[(u + 1) & 1][a] = *(const bf16x8*)(pa + (2 * s1 + a) * 2048 + (((4 * k1 + g4) ^ sw16) << 4));
        }
        if (u == 1) {
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) gb[1][bj] = *(const bf16x8*)(pb + bj * 2048 + (((4 + g4) ^ sw16) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc[bj][2 * sbk + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[kk][bj], ga[u & 1][a], acc[bj][2 * sbk + a], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }

    // Only DMA is outstanding here (the previous epilogue's stores retired before this tile's second K tile):
    // wait for the next tile's first K tile now, so the stores below never sit in front of a ring hand-off.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed = true;
    if (p.abl & 2) sink_acc(p, acc, tid);
    else store_direct<EPI, HAS_C2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
    sa = sa1;
    sb = sb1;
  }
}


// ------------------------------------------------------------------------------------------------
// gemm_nte = gemm_ntd with DEFERRED stores.  Measured on MI355X (profiles/r02_gemm_epilogue_experiments.md): a CU
// moves vector stores at ~16 B/clk, a wave that issues a store while that path is busy stalls AT ISSUE, and with all
// eight waves storing their 128 KiB tile at once the matrix pipe idles for ~4.3 us per tile (17 % of a K = 1024
// tile) - with no DMA wait involved at all.  The store path and the matrix pipe do run concurrently when the stores
// arrive slowly: so the epilogue only CONVERTS the tile (bias, activation / residual maths, bf16 packing: 64
// registers per lane) and the packed tile is stored two chunks per K tile during K tiles 1..8 of the NEXT output
// tile's main loop (16 KiB per CU and K tile: a third of the store path's capacity).  The DMA wait of K tile kt then
// allows the two stores issued after its DMA to stay in flight (`vmcnt(2)`); loads and stores retire in order, so a
// store has a full K tile to retire before it could hold up a ring hand-off.  The pre-activation copy (C2) of the
// two-output epilogue is still stored at once.  Needs K >= 640 (10 K tiles); the host routes smaller K to gemm_ntd.
template <int EPI, bool HAS_C2>
__global__ __launch_bounds__(NTHREADS) void gemm_nte_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  const int r8l = lane >> 3;
  const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  const int kel = chunk * 8;
  const int i16 = 8 * (wave & 1) + r8l, bjw = wave >> 1;
  const unsigned voffA = (unsigned)((wave * 8 + r8l) * p.lda * 2 + chunk * 16);
  const unsigned voffB = (unsigned)(bperm(16 * bjw + i16) * p.ldb * 2 + chunk * 16);
  const unsigned stepA = (unsigned)(64 * p.lda * 2), stepB = (unsigned)(64 * p.ldb * 2);
  const int nkt = (p.K + BK - 1) / BK;      // host guarantees nkt >= 10

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = (p.abl & 8) ? 1 : 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN, p.N - n) * p.ldb * 2)); };
  auto stage = [&](unsigned buf, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + kel >= p.K) ? 0x80000000u : 0u;
    const unsigned dA = lds0 + buf * STAGE_BYTES + wave * 1024, dB = dA + IMG_BYTES;
    // the per-piece row offsets are re-derived at every call (2 VALU ops per DMA) instead of living in 8 registers
    // across the main loop: the opaque copies keep hipcc from hoisting them (it would spill them to scratch, and a
    // scratch reload is a VMEM operation in the middle of the counted vmcnt protocol)
    unsigned qa = stepA, qb = stepB;
    asm volatile("" : "+s"(qa), "+s"(qb));
    dma16_x4<8192>(rsA, dA, voffA | oob, (voffA + qa) | oob, (voffA + 2 * qa) | oob, (voffA + 3 * qa) | oob, (unsigned)(k0 * 2));
    dma16_x4<8192>(rsB, dB, voffB | oob, (voffB + qb) | oob, (voffB + 2 * qb) | oob, (voffB + 3 * qb) | oob, (unsigned)(k0 * 2));
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  u32x4 sa = srdA(m0), sb = srdB(n0);
  stage(0, sa, sb, 0);
  unsigned gk = 0;
  bool landed = false;
  const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
  // the s = 0 half (chunk ai -> held[ai]: 32 registers) of the PREVIOUS output tile's packed result, and where it goes
  u32x4 held[8];
  bool have_held = false;
  __amdgpu_buffer_rsrc_t hrs = make_rsrc(p.C, 0);
  int hvoff = 0;
  bool hnok0 = false;
  const int stepC = (int)(16 * p.ldc * 2);
  auto put = [&](int i) {       // i is a compile-time constant at every call site
    if (hnok0) __builtin_amdgcn_raw_buffer_store_b128(held[i], hrs, hvoff + i * stepC, 0, 0);
  };
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);
    const u32x4 sa1 = srdA(m1), sb1 = srdB(n1);

    f32x4v acc[4][8];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      if (!(kt == 0 && landed)) {
        if (have_held && kt >= 2 && kt <= 9) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // newest = the store of K tile kt-1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      WG_BARRIER_LDS();
      if (kt + 1 < nkt) stage((gk + 1) & 1, sa, sb, (kt + 1) * BK);
      else if (has_next) stage((gk + 1) & 1, sa1, sb1, 0);
      if (have_held) {
        switch (kt) {
          case 1: put(0); break;
          case 2: put(1); break;
          case 3: put(2); break;
          case 4: put(3); break;
          case 5: put(4); break;
          case 6: put(5); break;
          case 7: put(6); break;
          case 8: put(7); break;
          default: break;
        }
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 64 + l15) * 128;
      // B fragments are single-buffered here (the packed tile of the previous output tile takes 64 registers): block
      // bj of k-step 1 is re-loaded right after the last MFMA pair that reads block bj of k-step 0
      bf16x8 ga[2][2], gb[4];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[bj] = *(const bf16x8*)(pb + bj * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int a = 0; a < 2; ++a) ga[0][a] = *(const bf16x8*)(pa + a * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sbk = u & 3;
        if (u < 7) {
          const int k1 = (u + 1) >> 2, s1 = (u + 1) & 3;
#pragma unroll
          for (int a = 0; a < 2; ++a) ga[(u + 1) & 1][a] = *(const bf16x8*)(pa + (2 * s1 + a) * 2048 + (((4 * k1 + g4) ^ sw16) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc[bj][2 * sbk + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[bj], ga[u & 1][a], acc[bj][2 * sbk + a], 0, 0, 0);
          if (u == 3) gb[bj] = *(const bf16x8*)(pb + bj * 2048 + (((4 + g4) ^ sw16) << 4));
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }

    // every deferred store of the previous tile was issued by K tile 8 and waited for by K tile 10
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed = true;
    if (p.abl & 2) {
      sink_acc(p, acc, tid);
#pragma unroll
      for (int i = 0; i < 8; ++i) held[i] = u32x4{0, 0, 0, 0};
      have_held = false;
    } else {
      const int mw = m0 + wm * 128, nw = n0 + wn * 64;
      epilogue_direct<EPI, HAS_C2, true>(p, acc, mw, nw, lane, held);
      const int rows = max(0, min(128, p.M - mw));
      hrs = make_rsrc(p.C + (size_t)mw * p.ldc * 2, (unsigned)((long)rows * p.ldc * 2));
      const int n = nw + 8 * g4;
      hvoff = (int)((l15 * p.ldc + n) * 2);
      hnok0 = n < p.N;
      have_held = true;
    }
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
    sa = sa1;
    sb = sb1;
  }
  if (have_held) {
#pragma unroll
    for (int i = 0; i < 8; ++i) put(i);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt6: 256 (m) x 128 (n) output tile, K step 32, FOUR waves (2 x 2, wave tile 128 x 64 as above), and TWO
// persistent workgroups per CU (72 KiB of LDS each, <= 256 registers per lane): the two workgroups are independent
// instruction streams on the same four SIMDs, so one workgroup's epilogue (conversion, activation, stores) and its
// ring hand-off run underneath the other's MFMA main loop instead of idling the matrix pipe.
//   LDS slot (24 KiB) = A image [256 rows][32 k] + B image [128 rows][32 k], 64-byte rows (4 chunks of 16 B);
//   physical chunk = logical chunk ^ ((-(row >> 2)) & 3): conflict-free for the 16x16x32 fragment read
//   (lane (c, g) -> row c, chunk g; tools/lds_bank_sim.py).  Ring of 3 slots, two K steps in flight, counted
//   `vmcnt(6)` (6 DMA instructions per wave and step).  The K steps 0 and 1 of the NEXT output tile are issued during
//   the last two steps of the current one and waited for at the top of the epilogue, before any store is issued;
//   steps 0 and 1 after an epilogue therefore need a barrier only, and the stores have until step 2 to drain.
constexpr int NT6_THREADS = 256, BN6 = 128;
constexpr int SLOT6 = (256 + 128) * 64;      // 24576
constexpr int LDS6_BYTES = 3 * SLOT6;        // 73728

template <int EPI, bool HAS_C2, int PRIO>
__device__ __forceinline__ void nt6_body(const NTArgs& p, char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tilesN = (p.N + BN6 - 1) / BN6;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  // DMA piece = 16 rows x 64 B; lane -> row (lane>>2), physical chunk lane&3.  Pieces start at multiples of 16 rows,
  // so the swizzle key (-(row>>2))&3 = (-(lane>>4))&3 is the same for every piece.
  const int r16 = lane >> 2;
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int kel = chunk * 8;
  // A pieces wave + 4j (j < 4): rows 64*j + 16*wave + r16.   B pieces wave + 4j (j < 2): image rows 64*j + 16*wave + r16
  // = MFMA row r16 of n block `wave` of wave column j  <-  feature 64*j + bperm(16*wave + r16)
  const unsigned voffA = (unsigned)((16 * wave + r16) * p.lda * 2 + chunk * 16);
  const unsigned voffB = (unsigned)(bperm(16 * wave + r16) * p.ldb * 2 + chunk * 16);
  const unsigned stepA = (unsigned)(64 * p.lda * 2), stepB = (unsigned)(64 * p.ldb * 2);
  const int nk = (p.K + 31) / 32;      // host guarantees nk >= 3

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = (p.abl & 8) ? 1 : 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN6;
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN6, p.N - n) * p.ldb * 2)); };
  auto stage = [&](unsigned slot, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + kel >= p.K) ? 0x80000000u : 0u;
    const unsigned d = lds0 + slot * SLOT6 + wave * 1024;
    dma16_x4<4096>(rsA, d, voffA | oob, (voffA + stepA) | oob, (voffA + 2 * stepA) | oob, (voffA + 3 * stepA) | oob, (unsigned)(k0 * 2));
    dma16_x2<4096>(rsB, d + 16384, voffB | oob, (voffB + stepB) | oob, (unsigned)(k0 * 2));
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  u32x4 sa = srdA(m0), sb = srdB(n0);
  stage(0, sa, sb, 0);
  stage(1, sa, sb, 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned cur = 0;       // ring slot of the step being computed; the refill target is (cur + 2) % 3
  // fragment read: lane (c, g) -> row c of a 16-row block, logical chunk g
  const int l15 = lane & 15, g4 = lane >> 4;
  const int coff = (g4 ^ ((0 - (l15 >> 2)) & 3)) << 4;
  const int offA = (wm * 128 + l15) * 64 + coff, offB = 16384 + (wn * 64 + l15) * 64 + coff;
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);
    const u32x4 sa1 = srdA(m1), sb1 = srdB(n1);

    f32x4v acc[4][8];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      // step t must have landed (steps 0, 1: waited for before the loop / at the top of the previous epilogue)
      if (t >= 2) {
        if (t + 1 < nk || has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // newest 6 = step t+1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      WG_BARRIER_LDS();
      const char* sl = smem + cur * SLOT6;
      bf16x8 gb[4], ga[2][2];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[bj] = *(const bf16x8*)(sl + offB + bj * 1024);
#pragma unroll
      for (int a = 0; a < 2; ++a) ga[0][a] = *(const bf16x8*)(sl + offA + a * 1024);
      {
        // refill the slot every wave finished reading at step t-1 with step t+2 (or step 0 / 1 of the next tile)
        const unsigned tgt = cur == 0 ? 2u : cur - 1u;     // (cur + 2) % 3
        if (t + 2 < nk) stage(tgt, sa, sb, (t + 2) * 32);
        else if (has_next) stage(tgt, sa1, sb1, (t + 2 - nk) * 32);
      }
#pragma unroll
      for (int sbk = 0; sbk < 4; ++sbk) {
        if (sbk < 3) {
#pragma unroll
          for (int a = 0; a < 2; ++a) ga[(sbk + 1) & 1][a] = *(const bf16x8*)(sl + offA + (2 * (sbk + 1) + a) * 1024);
        }
        __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc[bj][2 * sbk + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[bj], ga[sbk & 1][a], acc[bj][2 * sbk + a], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      cur = cur == 2 ? 0u : cur + 1u;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // steps 0 and 1 of the next tile have landed; queue empty
    if (p.abl & 2) sink_acc(p, acc, tid);
    else store_direct<EPI, HAS_C2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
    sa = sa1;
    sb = sb1;
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt7: the 256x256 tile / 8 waves / one persistent workgroup per CU of gemm_ntd on a DEEP ring: K is walked
// in steps of 32 through five 32 KiB slots (A image [256][32] + B image [256][32], 64-byte rows, the chunk swizzle of
// gemm_nt6), FOUR steps in flight (~3 us of look-ahead instead of one 64-wide K tile), waits are counted
// (`vmcnt(12)`: 4 DMA instructions per wave and step, the three newest steps may still be in flight).  Measured
// motivation: with the DMA switched off the main loop of gemm_ntd runs at 1.7 PF, with its 2-slot ring at 1.34 PF -
// the loop waits on L2 -> LDS latency, not on the matrix pipe.
//   Hand-off after an epilogue: steps 0..3 of the next tile are issued during the last four steps of the current
//   one; the top of the epilogue waits for steps 0 and 1 only (`vmcnt(8)`), then issues its S stores; steps 2 and 3
//   are waited for with `vmcnt(12 + S)` (they are OLDER than the stores, loads and stores retire in order), so the
//   stores have until step 4 to drain.  A tile with ragged edges issues an unknown number of stores: it waits for
//   everything (`vmcnt(0)`) at the top of its epilogue instead.
constexpr int SLOT7 = 32768, NS7 = 5;
constexpr int LDS7_BYTES = NS7 * SLOT7;      // 163840

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int EPI, bool HAS_C2>
__global__ __launch_bounds__(NTHREADS) void gemm_nt7_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int S = HAS_C2 ? 32 : 16;        // stores per lane of a full tile's epilogue
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  // DMA piece = 16 rows x 64 B; lane -> row (lane>>2), physical chunk lane&3; swizzle key (-(lane>>4))&3 for every piece.
  // Wave w moves A pieces w, w+8 (rows 16w + r16, +128) and B pieces w, w+8: B image row 128*j + 16*w + r16 = MFMA row
  // r16 of n block (w & 3) of wave column 2*j + (w >> 2)  <-  feature 128*j + 64*(w >> 2) + bperm(16*(w & 3) + r16)
  const int r16 = lane >> 2;
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int kel = chunk * 8;
  const unsigned voffA = (unsigned)((16 * wave + r16) * p.lda * 2 + chunk * 16);
  const unsigned voffB = (unsigned)((64 * (wave >> 2) + bperm(16 * (wave & 3) + r16)) * p.ldb * 2 + chunk * 16);
  const unsigned stepA = (unsigned)(128 * p.lda * 2), stepB = (unsigned)(128 * p.ldb * 2);
  const int nk = (p.K + 31) / 32;      // host guarantees nk >= 4

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = (p.abl & 8) ? 1 : 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN, p.N - n) * p.ldb * 2)); };
  auto stage = [&](unsigned slot, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + kel >= p.K) ? 0x80000000u : 0u;
    const unsigned d = lds0 + slot * SLOT7 + wave * 1024;
    dma16_x2<8192>(rsA, d, voffA | oob, (voffA + stepA) | oob, (unsigned)(k0 * 2));
    dma16_x2<8192>(rsB, d + 16384, voffB | oob, (voffB + stepB) | oob, (unsigned)(k0 * 2));
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  u32x4 sa = srdA(m0), sb = srdB(n0);
#pragma unroll
  for (int t = 0; t < 4; ++t) stage(t, sa, sb, t * 32);
  wait_vm<0>();
  int pre = 4;            // steps of this tile already known to have landed (this wave's pieces)
  unsigned cur = 0;       // ring slot of the step being computed; refill target = (cur + 4) % 5
  const int l15 = lane & 15, g4 = lane >> 4;
  const int coff = (g4 ^ ((0 - (l15 >> 2)) & 3)) << 4;
  const int offA = (wm * 128 + l15) * 64 + coff, offB = 16384 + (wn * 64 + l15) * 64 + coff;
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);
    const u32x4 sa1 = srdA(m1), sb1 = srdB(n1);

    f32x4v acc[4][8];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      if (t >= pre) {
        if (t < 4) wait_vm<12 + S>();        // steps 2, 3 after a full tile's epilogue: older than its S stores
        else {
          const int younger = has_next ? 3 : min(3, nk - 1 - t);   // issued steps newer than t
          if (younger == 3) wait_vm<12>();
          else if (younger == 2) wait_vm<8>();
          else if (younger == 1) wait_vm<4>();
          else wait_vm<0>();
        }
      }
      WG_BARRIER_LDS();
      const char* sl = smem + cur * SLOT7;
      bf16x8 gb[4], ga[2][2];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[bj] = *(const bf16x8*)(sl + offB + bj * 1024);
#pragma unroll
      for (int a = 0; a < 2; ++a) ga[0][a] = *(const bf16x8*)(sl + offA + a * 1024);
      {
        // refill the slot every wave finished reading at step t-1 with step t+4 (or step 0..3 of the next tile)
        const unsigned tgt = cur == 0 ? 4u : cur - 1u;     // (cur + 4) % 5
        if (t + 4 < nk) stage(tgt, sa, sb, (t + 4) * 32);
        else if (has_next) stage(tgt, sa1, sb1, (t + 4 - nk) * 32);
      }
#pragma unroll
      for (int sbk = 0; sbk < 4; ++sbk) {
        if (sbk < 3) {
#pragma unroll
          for (int a = 0; a < 2; ++a) ga[(sbk + 1) & 1][a] = *(const bf16x8*)(sl + offA + (2 * (sbk + 1) + a) * 1024);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc[bj][2 * sbk + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[bj], ga[sbk & 1][a], acc[bj][2 * sbk + a], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      cur = cur == 4 ? 0u : cur + 1u;
    }

    // In flight here (if there is a next tile): its steps 0..3, nothing else.
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N) && !(p.abl & 3);
    if (full && has_next) { wait_vm<8>(); pre = 2; }     // steps 0, 1 landed; 2, 3 stay in flight across the stores
    else { wait_vm<0>(); pre = 4; }
    if (p.abl & 2) sink_acc(p, acc, tid);
    else store_direct<EPI, HAS_C2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
    sa = sa1;
    sb = sb1;
  }
}

// Experiment flag 64: the workgroups of the upper half of the grid (the second resident workgroup of every CU under
// round-robin dispatch) issue their MFMA clusters at priority 3 instead of 1, so that two co-resident workgroups do
// not drift into phase (both in their epilogue at once).
template <int EPI, bool HAS_C2>
__global__ __launch_bounds__(NT6_THREADS, 2) void gemm_nt6_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((p.abl & 64) && blockIdx.x >= (gridDim.x >> 1)) nt6_body<EPI, HAS_C2, 3>(p, smem);
  else nt6_body<EPI, HAS_C2, 1>(p, smem);
}
