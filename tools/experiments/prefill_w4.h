// spatten_amd — the one-wave-per-SIMD flash kernel (round 3).  Included by prefill_attn.hip (inside
// namespace spatten, after the helpers it uses: FlashParams, Mfma, dma16, rope_pair, round2, logit_scale ...).
//
// STATUS: an EXPERIMENT, compiled only by tools/experiments/build_w4.sh (-DSPATTEN_WITH_W4_EXPERIMENT), selected by SPATTEN_PREFILL_W4=1.  Parity-green (tools/experiments/check_prefill_w4.py and the whole prefill /
// full-size / protocol suites with the switch on) and SLOWER than prefill_pp128_kernel: 616-642 against 740-766 TFLOP/s at
// q = N = 8192 (profiles/r03_prefill_w4_anatomy.txt) — with one wave per SIMD nothing hides that wave's own
// issue slots, and the reference's two roundings per logit are ~8 vector instructions per MFMA.
//
// Same arithmetic as prefill_pp128_kernel (modify_llama.py:92 rotary on the queries, :111-113 both roundings of a logit,
// :137 fp32 softmax, P rounded to the model dtype before P.V, deferred rescale) on a different machine shape:
//
//   workgroup = 4 waves = one 256-row query block, ONE wave per SIMD, each wave 64 query rows (two 32-row q-blocks) and the
//   whole 512-entry register file: O^T (2 x 4 x 16 = 128) and the rotated Q fragments (2 x 8 x 4 = 64) live in the
//   ACCUMULATOR file for the whole launch, named inside the inline-asm MFMAs (below); the scores S (2 x 2 x 16), P
//   (2 x 2 x 2 x 4 packed words) and the LDS operand ring in the arch VGPRs.  64-key tiles; per tile a wave issues 64 MFMAs in
//   two halves:
//       half A:  S0(t+1) = K(t+1) Q0^T ;  O0 += Vt(t) P0(t)      beside   softmax(S1(t)) -> P1(t)      + the 8 LDS-DMA instructions
//       half B:  S1(t+1) = K(t+1) Q1^T ;  O1 += Vt(t) P1(t)      beside   softmax(S0(t+1)) -> P0(t+1)    of stage t + 2
//   i.e. the two-wave ping-pong of prefill_pp128_kernel folded into ONE instruction stream: the softmax of one q-block runs as
//   a software pipeline over its 16 logit pairs, one stage per MFMA gap of the other q-block (source order pinned with
//   sched_barrier + input-only empty asm statements).
//   K/Vt tiles by LDS-DMA into a 3-deep ring (stage j = { K(j+1), Vt(j) }, issued two tiles ahead — always 8 instructions per
//   wave and stage, also past the last tile — ONE barrier per tile and a counted vmcnt(8): the wave never drains its DMA queue).
// The first Q.K^T + softmax (loop iteration t = -1), the tiles that straddle the causal diagonal (or the end of the keys) and the
// last tile run the same pieces one after the other (qk_seq / sm_seq / pv_seq, not interleaved).
//
// Serves: 16-bit dtypes, d = 128, no mask / stash / column importance / progressive quantisation / key split.

// The accumulator file is owned BY NAME: O^T of (q-block qb, 32-row block db) = a[64 qb + 16 db .. +15], the rotated Q
// fragment (qb, step kk) = a[192 + 32 qb + 4 kk .. +3]; a[128:191] are free.  (As "a" / "+a" operands of the asm statements the
// register allocator spilled the Q fragments to scratch — four reloads in front of every Q.K^T MFMA — and moved the O tuples
// through 64 copies per tile at the loop's back edge.)  Every asm statement of the kernel lists the whole file as clobbered,
// so no compiler value can live there across one of them.  Audit after a compiler change: the -save-temps listing must show
// no v_accvgpr_* outside ASMSTART / ASMEND and no scratch.
#define W4_ACLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define W4_MFMA_ASM(NAME)                                                                                              \
  template <int QLO> __device__ static inline void s_first(f32x16& c, const frag& a) {   /* S = A . Q        */           \
    asm volatile(NAME " %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(a), "i"(QLO), "i"(QLO + 3) : W4_ACLOB);                 \
  }                                                                                                                    \
  template <int QLO> __device__ static inline void s_next(f32x16& c, const frag& a) {    /* S += A . Q       */           \
    asm volatile(NAME " %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(a), "i"(QLO), "i"(QLO + 3) : W4_ACLOB);                 \
  }                                                                                                                    \
  template <int OLO> __device__ static inline void o_acc(const frag& a, const frag& p) { /* O += A . P       */           \
    asm volatile(NAME " a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(p), "i"(OLO), "i"(OLO + 15) : W4_ACLOB);        \
  }
template <typename T> struct MfmaAsm;
template <> struct MfmaAsm<bf16_t> {
  using frag = typename Mfma<bf16_t>::frag;
  W4_MFMA_ASM("v_mfma_f32_32x32x16_bf16")
};
template <> struct MfmaAsm<f16_t> {
  using frag = typename Mfma<f16_t>::frag;
  W4_MFMA_ASM("v_mfma_f32_32x32x16_f16")
};
template <int REG> __device__ inline void w4_acc_write(uint32_t w) {
  asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(w), "i"(REG) : W4_ACLOB);
}
template <int REG> __device__ inline void w4_acc_zero() {
  asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(REG) : W4_ACLOB);
}
template <int REG> __device__ inline float w4_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(REG) : W4_ACLOB);
  return x;
}
template <int REG> __device__ inline void w4_acc_scale(float alpha) {     // a[REG] *= alpha
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %1, %0\n\tv_accvgpr_write_b32 a%c2, %0"
               : "=&v"(t) : "v"(alpha), "i"(REG) : W4_ACLOB);
}
template <int LO, int N> __device__ inline void w4_acc_scale_range(float alpha) {
  if constexpr (N == 1) w4_acc_scale<LO>(alpha);
  else { w4_acc_scale_range<LO, N / 2>(alpha); w4_acc_scale_range<LO + N / 2, N - N / 2>(alpha); }
}
template <int LO, int N> __device__ inline void w4_acc_zero_range() {
  if constexpr (N == 1) w4_acc_zero<LO>();
  else { w4_acc_zero_range<LO, N / 2>(); w4_acc_zero_range<LO + N / 2, N - N / 2>(); }
}

template <int I> struct IC { static constexpr int v = I; };

#ifdef SPATTEN_PF_TRACE   // tools/probe_w4_trace.py: shader-cycle stamps of workgroup 0, tiles 40..55
#ifndef SPATTEN_W4_STAMPS       // 1: only the tile-to-tile stamp (a stamp costs ~500 cycles itself)
#define SPATTEN_W4_STAMPS 0
#endif
#define W4_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if ((!SPATTEN_W4_STAMPS || (slot) == 0) && g_pf_trace && blockIdx.x == 0 && lane == 0 && t >= 40 && t < 56) \
      g_pf_trace[((wave_u * 16 + (t - 40)) * 8) + (slot)] = __builtin_readcyclecounter();                \
  } while (0)
#else
#define W4_STAMP(slot)
#endif

#ifndef SPATTEN_W4_EXP            // anatomy builds (tools/mb/w4_exp.sh; WRONG results): 1 exp2 -> multiply, 2 no softmax slices in the
#define SPATTEN_W4_EXP 0          // interleaved halves, 4 operand fragments fetched once per half, 8 no DMA in the tile loop,
#endif                            // 16 no barrier in the tile loop
#ifndef SPATTEN_W4_RING
#define SPATTEN_W4_RING 4          // LDS operand fragments in flight
#endif

template <typename T, bool FASTN>
__global__ __launch_bounds__(256, 1) void prefill_w4_kernel(const FlashParams<T> p) {
  constexpr int D = 128, KT = 64, KK = D / 16, DB = D / 32;
  constexpr int KBYTES = KT * 256, VBYTES = D * 128, BUF = KBYTES + VBYTES, NST = 3;   // 16 + 16 KB per stage, 3 stages
  constexpr int RING = SPATTEN_W4_RING;
  using frag = typename Mfma<T>::frag;
  using MA = MfmaAsm<T>;
  __shared__ __attribute__((aligned(1024))) char lds[NST * BUF];

  const int tid = threadIdx.x, lane = tid & 63, qi = lane & 31, hi = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = p.nqb;
  int h, qblk, b;
  {
    const int i = (int)blockIdx.x;
    const int per_b = p.H * nqb;
    b = i / per_b;
    const int j = i - b * per_b;
    if ((p.H & 7) == 0) {
      const int xx = j & 7, ss = j >> 3;
      h = xx + 8 * (ss / nqb);
      qblk = nqb - 1 - (ss % nqb);
    } else {
      h = j / nqb;
      qblk = nqb - 1 - (j % nqb);
    }
  }
  const int hkv = p.Hkv == p.H ? h : h / (p.H / p.Hkv);
  const int r0 = qblk * 256 + wave_u * 64;           // this wave's first query row
  const int P = p.N - p.q_len;
  const float rsqrt_d = 1.0f / p.sqrt_d;

  // ---- tiles ------------------------------------------------------------------------------------------------------
  const int wg_q_end = min(p.q_len, qblk * 256 + 256);
  const int att_keys = p.causal ? min(p.N, P + wg_q_end) : p.N;
  const int n_tiles = (att_keys + KT - 1) / KT;                                   // what the workgroup brings in
  const int wave_keys = r0 >= p.q_len ? 0 : (p.causal ? min(p.N, P + min(p.q_len, r0 + 64)) : p.N);
  const int wave_tiles = (wave_keys + KT - 1) / KT;                               // what this wave computes on
  // tiles below nfull need no visibility test for either q-block (q-block 0's first row sees all their keys)
  const int nfull = r0 >= p.q_len ? 0 : (p.causal ? min(p.N, P + r0 + 1) : p.N) / KT;

  // ---- Q fragments, rotated here (as prefill_pp128_kernel) ------------------------------------------------------
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qq = min(r0 + 32 * qb + qi, p.q_len - 1);
    const T* qrow = p.q + b * p.q_sb + h * p.q_sh + (int64_t)qq * p.q_sq;
    int ps = p.pos_ids ? (int)p.pos_ids[b * p.pos_sb + qq] : p.pos_q0 + qq;
    ps = min(max(ps, 0), p.table_rows - 1);
    const T* cr = p.cos + (int64_t)ps * (D / 2);
    const T* sr = p.sin + (int64_t)ps * (D / 2);
    auto park = [&](auto Qc, auto Kc, const frag& f) __attribute__((always_inline)) {
      constexpr int base = 192 + 32 * decltype(Qc)::v + 4 * decltype(Kc)::v;
      const u32x4 w = *reinterpret_cast<const u32x4*>(&f);
      w4_acc_write<base>(w[0]); w4_acc_write<base + 1>(w[1]); w4_acc_write<base + 2>(w[2]); w4_acc_write<base + 3>(w[3]);
    };
    auto rot = [&](auto Qc, auto Kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(Kc)::v;
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + 16 * kk + 8 * hi), xlo);
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + D / 2 + 16 * kk + 8 * hi), xhi);
      Vec8<T>::unpack(Vec8<T>::ldg(cr + 16 * kk + 8 * hi), cc);
      Vec8<T>::unpack(Vec8<T>::ldg(sr + 16 * kk + 8 * hi), ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
      frag flo, fhi;
#pragma unroll
      for (int e = 0; e < 8; ++e) { flo[e] = DT<T>::from_f32(ylo[e]); fhi[e] = DT<T>::from_f32(yhi[e]); }
      park(Qc, Kc, flo);
      park(Qc, IC<kk + KK / 2>{}, fhi);
    };
    if (qb == 0) { rot(IC<0>{}, IC<0>{}); rot(IC<0>{}, IC<1>{}); rot(IC<0>{}, IC<2>{}); rot(IC<0>{}, IC<3>{}); }
    else { rot(IC<1>{}, IC<0>{}); rot(IC<1>{}, IC<1>{}); rot(IC<1>{}, IC<2>{}); rot(IC<1>{}, IC<3>{}); }
  }
  w4_acc_zero_range<0, 128>();
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const int my_vis[2] = {p.causal ? min(p.N, P + r0 + qi + 1) : p.N, p.causal ? min(p.N, P + r0 + 32 + qi + 1) : p.N};

  const T* krb = p.kr + b * p.kv_sb + hkv * p.kv_sh;
  const T* vtb = p.vt + ((int64_t)(b * p.Hkv + hkv) * D) * p.Npad;
  const int64_t k_bytes = (int64_t)p.N * D * 2, v_bytes = (int64_t)D * p.Npad * 2;
  // stage j = { K(j+1), Vt(j) } in slot j mod 3 (K(0) is parked in the slot of stage -1)
  auto k_area = [&](int stage) __attribute__((always_inline)) -> char* { return lds + ((stage + NST) % NST) * BUF; };
  auto v_area = [&](int stage) __attribute__((always_inline)) -> char* { return lds + ((stage + NST) % NST) * BUF + KBYTES; };
  // K tile: 64 rows of 256 B, 16 one-KiB pieces of 4 rows, 4 per wave; 16-byte slot s of row r at physical slot s ^ (r & 15)
  auto dma_k = [&](int tile, char* area) __attribute__((always_inline)) {
    const int ln = opaque_lane(lane);
    const int lr = ln >> 4, ps = ln & 15;
    const int base = lr * 256 + ((ps ^ lr) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave_u * 4 + i;
      dma16(krb, k_bytes, area + piece * 1024, base ^ (((piece * 4) & 15) << 4), tile * (KT * 256) + piece * 1024);
    }
  };
  // Vt tile: 128 rows (dv) of 128 B (64 keys), 16 pieces of 8 rows, 4 per wave; slot s of row r at physical slot s ^ ((r >> 1) & 7)
  auto dma_v = [&](int tile, char* area) __attribute__((always_inline)) {
    const int ln = opaque_lane(lane);
    const int lr = ln >> 3, ps = ln & 7;
    const int base = lr * p.Npad * 2 + ((ps ^ (lr >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave_u * 4 + i;
      dma16(vtb, v_bytes, area + piece * 1024, base ^ ((piece & 1) << 6), tile * (KT * 2) + piece * 8 * p.Npad * 2);
    }
  };
  // A stage is ALWAYS 8 instructions per wave (K(j+1): 4, Vt(j): 4), also past the last tile — rows past N read as zeros,
  // a Vt tile past the padded length reads whatever follows inside the buffer, and either lands in a slot nobody reads — so
  // the loop's counted wait is always vmcnt(8).
  auto dma_stage = [&](int j) __attribute__((always_inline)) {
    dma_k(j + 1, k_area(j));
    dma_v(j, v_area(j));
  };
  // the same stage one instruction at a time (n = 0..3: K pieces, 4..7: Vt pieces), for the MFMA gaps of half A; the lane
  // terms (kl, vl: dma_lane) are computed once per half
  auto dma_lane = [&](int& kl, int& vl) __attribute__((always_inline)) {
    const int ln = opaque_lane(lane);
    kl = (ln >> 4) * 256 + (((ln & 15) ^ (ln >> 4)) << 4);
    vl = (ln >> 3) * p.Npad * 2 + (((ln & 7) ^ (ln >> 4)) << 4);
  };
  auto dma_piece = [&](auto Nc, int j, int kl, int vl) __attribute__((always_inline)) {
    constexpr int n = decltype(Nc)::v;
    if constexpr (n < 4) {
      const int piece = wave_u * 4 + n;
      dma16(krb, k_bytes, k_area(j) + piece * 1024, kl ^ (((piece * 4) & 15) << 4), (j + 1) * (KT * 256) + piece * 1024);
    } else {
      const int piece = wave_u * 4 + (n - 4);
      dma16(vtb, v_bytes, v_area(j) + piece * 1024, vl ^ ((piece & 1) << 6), j * (KT * 2) + piece * 8 * p.Npad * 2);
    }
  };

  f32x16 s[2][2];          // [q-block][32-key block]
  uint32_t pw[2][2][2][4]; // P in the model dtype, packed pairs: [q-block][32-key block][16-key step][word] = one MFMA operand

  // ---- operand fragments -----------------------------------------------------------------------------------------
  // step i of a half: 0..15 = Q.K^T (kk = i >> 1, kb = i & 1), 16..31 = P.V (kt = (i - 16) >> 2, db = (i - 16) & 3)
  // (ku / vu already carry the lane's hi slot bit)
  auto kfrag = [&](unsigned ku, int i) __attribute__((always_inline)) {
    return *reinterpret_cast<const frag*>(lds + ((ku ^ ((2 * (i >> 1)) << 4)) + (i & 1) * 32 * 256));
  };
  auto vfrag = [&](unsigned vu, int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const frag*>(lds + ((vu ^ ((2 * (j >> 2)) << 4)) + (j & 3) * 32 * 128));
  };
  // (recomputed from an opaque lane id wherever they are used: hoisted out of the tile loop, the per-stage variants are
  //  spilled, and a reload's vmcnt(0) drains the DMA queue)
  auto k_base = [&](const char* kbuf) __attribute__((always_inline)) {
    const int q = opaque_lane(lane) & 31;
    return ((unsigned)(kbuf - lds) + q * 256 + ((q & 15) << 4)) ^ ((unsigned)(opaque_lane(lane) >> 5) << 4);
  };
  auto v_base = [&](const char* vbuf) __attribute__((always_inline)) {
    const int q = opaque_lane(lane) & 31;
    return ((unsigned)(vbuf - lds) + q * 128 + (((q >> 1) & 7) << 4)) ^ ((unsigned)(opaque_lane(lane) >> 5) << 4);
  };

  auto mma_step = [&](auto Mc, auto Ic, const frag& a) __attribute__((always_inline)) {
    constexpr int M = decltype(Mc)::v, i = decltype(Ic)::v;
    if constexpr (i < 16) {
      constexpr int qlo = 192 + 32 * M + 4 * (i >> 1);
      if constexpr ((i >> 1) == 0) MA::template s_first<qlo>(s[M][i & 1], a);
      else MA::template s_next<qlo>(s[M][i & 1], a);
    } else {
      constexpr int j = i - 16, kt = j >> 2, db = j & 3;
      const u32x4 u = {pw[M][kt >> 1][kt & 1][0], pw[M][kt >> 1][kt & 1][1], pw[M][kt >> 1][kt & 1][2], pw[M][kt >> 1][kt & 1][3]};
      MA::template o_acc<64 * M + 16 * db>(a, __builtin_bit_cast(frag, u));
    }
  };

  // ---- softmax of q-block V, tile `tile`, in chunks ----------------------------------------------------------------
  // chunks 0..15: both roundings + running max of logit pair c; chunk 16: the row maximum, the deferred-rescale decision
  // (and the rescale); chunks 17..32: exp / sum / P of pair c - 17.
  constexpr int NCH = 33;
  float mt[2], ls[4], m2 = 0.f, x[32];   // x: the 32 (rounded, masked) logits of the q-block in hand
  auto sm_chunk = [&](auto Vc, auto Cc, auto Ec, int tile) __attribute__((always_inline)) {
    constexpr int V = decltype(Vc)::v, c = decltype(Cc)::v;
    constexpr bool EDGE = decltype(Ec)::v != 0;
    if constexpr (c < 16) {
      constexpr int kb = c >> 3, r = 2 * (c & 7);
      float v0 = s[V][kb][r], v1 = s[V][kb][r + 1];
      if constexpr (!FASTN) {
        const f32x2 x = round2<T>(f32x2{v0, v1});
        const f32x2 y = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
        v0 = y[0]; v1 = y[1];
      } else if constexpr (EDGE) {
        v0 *= rsqrt_d; v1 *= rsqrt_d;
      }
      if constexpr (EDGE) {
        const int key = tile * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;      // r is even: r + 1 is key + 1
        v0 = key < my_vis[V] ? v0 : -INFINITY;
        v1 = key + 1 < my_vis[V] ? v1 : -INFINITY;
      }
      x[2 * c] = v0; x[2 * c + 1] = v1;
      if constexpr (c < 2) mt[c & 1] = fmaxf(v0, v1);
      else mt[c & 1] = max3_raw(mt[c & 1], v0, v1);
    } else if constexpr (c == 16) {
      float m_tile = xor32_max(fmaxf(mt[0], mt[1]));
      if constexpr (FASTN && !EDGE) m_tile *= rsqrt_d;
      // the rescale is taken by the WHOLE wave (a scalar branch; rows whose maximum stays get alpha = 1, exact): a per-lane
      // branch makes the compiler merge the accumulator tuples through 64 arch VGPRs on the common path
      float m_new, m_base;
      bool any;
      if constexpr (!EDGE) {
        any = __builtin_amdgcn_ballot_w64(m_tile - m_run[V] > kDeferMax) != 0;   // -inf start: inf > thr
        m_new = any ? fmaxf(m_run[V], m_tile) : m_run[V];
        m_base = m_new;
      } else {
        m_new = fmaxf(m_run[V], m_tile);
        any = __builtin_amdgcn_ballot_w64(m_new != m_run[V]) != 0;
        m_base = (m_new == -INFINITY) ? 0.f : m_new;       // nothing visible yet: exp2(-inf) = 0 for every key
      }
      if (any) {
        const float alpha = (m_new == m_run[V]) ? 1.f : __expf(m_run[V] - m_base);
        l_run[V] *= alpha;
        w4_acc_scale_range<64 * V, 64>(alpha);
        m_run[V] = m_new;
      }
      m2 = m_base * kLog2e;
      ls[0] = ls[1] = ls[2] = ls[3] = 0.f;
    } else {
      constexpr int pr = c - 17, kb = pr >> 3, r = 2 * (pr & 7);
      const float sc2 = (FASTN && !EDGE) ? kLog2e * rsqrt_d : kLog2e;
#if SPATTEN_W4_EXP & 1
      const float p0 = fmaf(x[2 * pr], sc2, -m2) * sc2, p1 = fmaf(x[2 * pr + 1], sc2, -m2) * sc2;
#else
      const float p0 = __builtin_amdgcn_exp2f(fmaf(x[2 * pr], sc2, -m2));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(x[2 * pr + 1], sc2, -m2));
#endif
      ls[pr & 3] += p0;
      ls[(pr + 2) & 3] += p1;
      // (pinned here with empty asm statements: the branch of chunk 16 ends the basic block, and LLVM otherwise SINKS these
      //  pure computations into the block of their first use — the P.V of the NEXT half — leaving the MFMA gaps empty)
      typedef T t2 __attribute__((ext_vector_type(2)));
      const t2 pk = {DT<T>::from_f32(p0), DT<T>::from_f32(p1)};
      uint32_t w = __builtin_bit_cast(uint32_t, pk);
      asm volatile("" : : "v"(w), "v"(ls[pr & 3]), "v"(ls[(pr + 2) & 3]));
      pw[V][kb][r >> 3][(r & 7) >> 1] = w;
      if constexpr (c == NCH - 1) { l_run[V] += (ls[0] + ls[1]) + (ls[2] + ls[3]); asm volatile("" : : "v"(l_run[V])); }
    }
  };
  // The same softmax for the MFMA gaps of a FULL tile (no visibility test), as a SOFTWARE PIPELINE over the 16 logit pairs of
  // the q-block: one wave has nobody to hide a dependent instruction behind, so a gap never holds two consecutive stages of
  // the same pair — pair p enters at gap BASE + p (32 - BASE - NSTG) / 15 and moves one stage per gap:
  //   reference numerics (6 stages, BASE 0):  round(acc) | x / sqrt(d) | round | x log2e - m | exp2 | row sums + P word
  //   fast numerics      (3 stages, BASE 3):  acc log2e / sqrt(d) - m | exp2 | row sums + P word
  // Gaps 0, 1: the maximum of the 32 RAW accumulators (both roundings and the scale are monotone: the maximum of the rounded
  // logits is the rounded maximum), two interleaved max3 chains; gap 2: lane <-> lane + 32, the reference roundings of that
  // one value, the deferred-rescale decision and the rescale.  ~8 (reference) / ~4 (fast) vector instructions per gap.
  // Every stage ends on an empty asm statement over what it produced: pure instructions are ordered against sched_barrier
  // only through such a statement (input-only: an output operand would cost an s_nop after it), and LLVM otherwise sinks
  // them into the block of their first use, the P.V of the next half.
  constexpr int NSTG = FASTN ? 3 : 6, BASE = FASTN ? 3 : 0;
  f32x2 ya[16], fd[16];
  auto sm_max = [&](auto Vc, auto Cc) __attribute__((always_inline)) {
    constexpr int V = decltype(Vc)::v, c = decltype(Cc)::v;
    if constexpr (c < 2) {
      constexpr int r0_ = 8 * c;
      if constexpr (c == 0) { mt[0] = fmaxf(s[V][0][0], s[V][0][1]); mt[1] = fmaxf(s[V][1][0], s[V][1][1]); }
      else { mt[0] = max3_raw(mt[0], s[V][0][8], s[V][0][9]); mt[1] = max3_raw(mt[1], s[V][1][8], s[V][1][9]); }
      mt[0] = max3_raw(mt[0], s[V][0][r0_ + 2], s[V][0][r0_ + 3]); mt[1] = max3_raw(mt[1], s[V][1][r0_ + 2], s[V][1][r0_ + 3]);
      mt[0] = max3_raw(mt[0], s[V][0][r0_ + 4], s[V][0][r0_ + 5]); mt[1] = max3_raw(mt[1], s[V][1][r0_ + 4], s[V][1][r0_ + 5]);
      mt[0] = max3_raw(mt[0], s[V][0][r0_ + 6], s[V][0][r0_ + 7]); mt[1] = max3_raw(mt[1], s[V][1][r0_ + 6], s[V][1][r0_ + 7]);
      asm volatile("" : : "v"(mt[0]), "v"(mt[1]));
    } else {
      float m_tile = xor32_max(fmaxf(mt[0], mt[1]));
      if constexpr (FASTN) m_tile *= rsqrt_d;
      else m_tile = DT<T>::round(logit_scale<T>(DT<T>::round(m_tile), p.sqrt_d, rsqrt_d));
      const bool any = __builtin_amdgcn_ballot_w64(m_tile - m_run[V] > kDeferMax) != 0;   // -inf start: inf > thr
      const float m_new = any ? fmaxf(m_run[V], m_tile) : m_run[V];
      if (any) {       // a scalar branch: rows whose maximum stays get alpha = 1 (exact)
        const float alpha = (m_new == m_run[V]) ? 1.f : __expf(m_run[V] - m_new);
        l_run[V] *= alpha;
        w4_acc_scale_range<64 * V, 64>(alpha);
        m_run[V] = m_new;
      }
      m2 = m_new * kLog2e;
      ls[0] = ls[1] = ls[2] = ls[3] = 0.f;
    }
  };
  auto sm_stage = [&](auto Vc, auto Pc, auto Sc) __attribute__((always_inline)) {
    constexpr int V = decltype(Vc)::v, pr = decltype(Pc)::v, st = decltype(Sc)::v + (FASTN ? 3 : 0);
    constexpr int kb = pr >> 3, r = 2 * (pr & 7);
    if constexpr (st == 0) {
      ya[pr] = round2<T>(f32x2{s[V][kb][r], s[V][kb][r + 1]});
      asm volatile("" : : "v"(ya[pr][0]), "v"(ya[pr][1]));
    } else if constexpr (st == 1) {
      ya[pr] = f32x2{logit_scale<T>(ya[pr][0], p.sqrt_d, rsqrt_d), logit_scale<T>(ya[pr][1], p.sqrt_d, rsqrt_d)};
      asm volatile("" : : "v"(ya[pr][0]), "v"(ya[pr][1]));
    } else if constexpr (st == 2) {
      ya[pr] = round2<T>(ya[pr]);
      asm volatile("" : : "v"(ya[pr][0]), "v"(ya[pr][1]));
    } else if constexpr (st == 3) {
      if constexpr (FASTN) fd[pr] = f32x2{fmaf(s[V][kb][r], kLog2e * rsqrt_d, -m2), fmaf(s[V][kb][r + 1], kLog2e * rsqrt_d, -m2)};
      else fd[pr] = f32x2{fmaf(ya[pr][0], kLog2e, -m2), fmaf(ya[pr][1], kLog2e, -m2)};
      asm volatile("" : : "v"(fd[pr][0]), "v"(fd[pr][1]));
    } else if constexpr (st == 4) {
#if SPATTEN_W4_EXP & 1
      fd[pr] = f32x2{fd[pr][0] * kLog2e, fd[pr][1] * kLog2e};
#else
      fd[pr] = f32x2{__builtin_amdgcn_exp2f(fd[pr][0]), __builtin_amdgcn_exp2f(fd[pr][1])};
#endif
      asm volatile("" : : "v"(fd[pr][0]), "v"(fd[pr][1]));
    } else {
      ls[pr & 3] += fd[pr][0];
      ls[(pr + 2) & 3] += fd[pr][1];
      typedef T t2 __attribute__((ext_vector_type(2)));
      const t2 pk = {DT<T>::from_f32(fd[pr][0]), DT<T>::from_f32(fd[pr][1])};
      const uint32_t w = __builtin_bit_cast(uint32_t, pk);
      asm volatile("" : : "v"(w), "v"(ls[pr & 3]), "v"(ls[(pr + 2) & 3]));
      pw[V][kb][r >> 3][(r & 7) >> 1] = w;
      if constexpr (pr == 15) { l_run[V] += (ls[0] + ls[1]) + (ls[2] + ls[3]); asm volatile("" : : "v"(l_run[V])); }
    }
  };
  auto sm_slice = [&](auto Vc, auto Ic) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::v;
    if constexpr (i < 3) sm_max(Vc, Ic);
    auto one = [&](auto Pc) __attribute__((always_inline)) {
      constexpr int pr = decltype(Pc)::v, st = i - BASE - pr * (32 - BASE - NSTG) / 15;
      if constexpr (st >= 0 && st < NSTG) sm_stage(Vc, Pc, IC<st>{});
    };
    one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{}); one(IC<4>{}); one(IC<5>{}); one(IC<6>{}); one(IC<7>{});
    one(IC<8>{}); one(IC<9>{}); one(IC<10>{}); one(IC<11>{}); one(IC<12>{}); one(IC<13>{}); one(IC<14>{}); one(IC<15>{});
  };

  // ---- the interleaved half: 32 MFMAs of q-block M, slice i of softmax(V) behind MFMA i ------------------------------
  auto half = [&](auto Mc, auto Vc, const char* kbuf, const char* vbuf, int dma_j) __attribute__((always_inline)) {
    const unsigned ku = k_base(kbuf), vu = v_base(vbuf);
    int kl = 0, vl = 0;
    if constexpr (decltype(Mc)::v == 0) dma_lane(kl, vl);
    frag a[RING];
    auto fetch = [&](int i) __attribute__((always_inline)) { return i < 16 ? kfrag(ku, i) : vfrag(vu, i - 16); };
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = fetch(i);
    __builtin_amdgcn_sched_barrier(0);
    auto body = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::v;
      mma_step(Mc, Ic, a[i % RING]);
      if constexpr (i + RING < 32 && !(SPATTEN_W4_EXP & 4)) a[i % RING] = fetch(i + RING);
      // half A carries the 8 LDS-DMA instructions of stage t + 2, one behind every other MFMA (issued as a burst at the top of
      // the tile they cost ~1,100 cycles with the matrix pipe idle)
      if constexpr (decltype(Mc)::v == 0 && (i & 1) == 0 && i < 16 && !(SPATTEN_W4_EXP & 8)) dma_piece(IC<(i >> 1)>{}, dma_j, kl, vl);
      if constexpr (!(SPATTEN_W4_EXP & 2)) sm_slice(Vc, Ic);
      __builtin_amdgcn_sched_barrier(0);
    };
    body(IC<0>{}); body(IC<1>{}); body(IC<2>{}); body(IC<3>{}); body(IC<4>{}); body(IC<5>{}); body(IC<6>{}); body(IC<7>{});
    body(IC<8>{}); body(IC<9>{}); body(IC<10>{}); body(IC<11>{}); body(IC<12>{}); body(IC<13>{}); body(IC<14>{}); body(IC<15>{});
    body(IC<16>{}); body(IC<17>{}); body(IC<18>{}); body(IC<19>{}); body(IC<20>{}); body(IC<21>{}); body(IC<22>{}); body(IC<23>{});
    body(IC<24>{}); body(IC<25>{}); body(IC<26>{}); body(IC<27>{}); body(IC<28>{}); body(IC<29>{}); body(IC<30>{}); body(IC<31>{});
  };

  // ---- the same pieces one after the other (first / last / diagonal tiles) --------------------------------------------
  auto qk_seq = [&](auto Mc, const char* kbuf) __attribute__((always_inline)) {
    const unsigned ku = k_base(kbuf);
    frag a[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = kfrag(ku, i);
    auto body = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::v;
      mma_step(Mc, Ic, a[i % RING]);
      if constexpr (i + RING < 16) a[i % RING] = kfrag(ku, i + RING);
    };
    body(IC<0>{}); body(IC<1>{}); body(IC<2>{}); body(IC<3>{}); body(IC<4>{}); body(IC<5>{}); body(IC<6>{}); body(IC<7>{});
    body(IC<8>{}); body(IC<9>{}); body(IC<10>{}); body(IC<11>{}); body(IC<12>{}); body(IC<13>{}); body(IC<14>{}); body(IC<15>{});
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // the last MFMA's result registers -> the softmax's VALU reads
    __builtin_amdgcn_sched_barrier(0);
  };
  auto pv_seq = [&](auto Mc, const char* vbuf) __attribute__((always_inline)) {
    const unsigned vu = v_base(vbuf);
    frag a[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = vfrag(vu, i);
    __builtin_amdgcn_sched_barrier(0);
    auto body = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::v;
      mma_step(Mc, IC<i + 16>{}, a[i % RING]);
      if constexpr (i + RING < 16) a[i % RING] = vfrag(vu, i + RING);
    };
    body(IC<0>{}); body(IC<1>{}); body(IC<2>{}); body(IC<3>{}); body(IC<4>{}); body(IC<5>{}); body(IC<6>{}); body(IC<7>{});
    body(IC<8>{}); body(IC<9>{}); body(IC<10>{}); body(IC<11>{}); body(IC<12>{}); body(IC<13>{}); body(IC<14>{}); body(IC<15>{});
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // ... -> a rescale's accumulator reads
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sm_seq_e = [&](auto Vc, auto Ec, int tile) __attribute__((always_inline)) {
    auto ch = [&](auto Cc) __attribute__((always_inline)) { sm_chunk(Vc, Cc, Ec, tile); };
    ch(IC<0>{}); ch(IC<1>{}); ch(IC<2>{}); ch(IC<3>{}); ch(IC<4>{}); ch(IC<5>{}); ch(IC<6>{}); ch(IC<7>{});
    ch(IC<8>{}); ch(IC<9>{}); ch(IC<10>{}); ch(IC<11>{}); ch(IC<12>{}); ch(IC<13>{}); ch(IC<14>{}); ch(IC<15>{});
    ch(IC<16>{});
    ch(IC<17>{}); ch(IC<18>{}); ch(IC<19>{}); ch(IC<20>{}); ch(IC<21>{}); ch(IC<22>{}); ch(IC<23>{}); ch(IC<24>{});
    ch(IC<25>{}); ch(IC<26>{}); ch(IC<27>{}); ch(IC<28>{}); ch(IC<29>{}); ch(IC<30>{}); ch(IC<31>{}); ch(IC<32>{});
    asm volatile("s_nop 1" ::: "memory");                 // the last P conversion -> an MFMA's operand read
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sm_seq = [&](auto Vc, int tile) __attribute__((always_inline)) {
    if (tile < nfull) sm_seq_e(Vc, IC<0>{}, tile);
    else sm_seq_e(Vc, IC<1>{}, tile);
  };

  // ---- prologue: K(0) in the slot of stage -1, stage 0; the loop starts at t = -1 (the first Q.K^T and softmax) ----------
  dma_k(0, k_area(-1));
  dma_stage(0);

  for (int t = -1; t < n_tiles; ++t) {
    // stage t has landed (this wave's pieces: everything but the 8 instructions of stage t + 1, issued one iteration ago);
    // after the barrier everyone is also done with stage t - 1, whose slot takes stage t + 2
    W4_STAMP(0);
    __builtin_amdgcn_s_waitcnt(0x0F78);        // vmcnt(8)
    if (!(SPATTEN_W4_EXP & 16) || t < 2) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4_STAMP(1);
    const bool steady = t >= 0 && t + 1 < nfull;            // (nfull <= wave_tiles)
    if (steady) {
      half(IC<0>{}, IC<1>{}, k_area(t), v_area(t), t + 2);
      W4_STAMP(3);
      half(IC<1>{}, IC<0>{}, k_area(t), v_area(t), 0);
      W4_STAMP(4);
    } else {
      if (!(SPATTEN_W4_EXP & 8) || t < 2) dma_stage(t + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (t < wave_tiles) {
        const bool more = t + 1 < wave_tiles, live = t >= 0;
        if (more) qk_seq(IC<0>{}, k_area(t));
        if (live) { pv_seq(IC<0>{}, v_area(t)); sm_seq(IC<1>{}, t); }
        if (more) qk_seq(IC<1>{}, k_area(t));
        if (live) pv_seq(IC<1>{}, v_area(t));
        if (more) sm_seq(IC<0>{}, t + 1);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);          // no LDS-DMA may outlive the workgroup
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");

  // ---- epilogue: O = O^T / l -------------------------------------------------------------------------------------------
  auto store_q = [&](auto Qc) __attribute__((always_inline)) {
    constexpr int qb = decltype(Qc)::v;
    const int myq = r0 + 32 * qb + qi;
    const bool qvalid = myq < p.q_len;
    const float l_tot = xor32_sum(l_run[qb]);
    const float inv = 1.f / l_tot;
    if (p.lse != nullptr && qvalid && hi == 0) {
      float* lo = p.lse + ((int64_t)(b * p.H + h) * p.q_len + myq) * 2;
      lo[0] = m_run[qb]; lo[1] = l_tot;
    }
    T* orow = p.out + b * p.out_sb + (int64_t)min(myq, p.q_len - 1) * p.out_sq + h * D;
    auto store4 = [&](auto Rc) __attribute__((always_inline)) {      // registers 4g .. 4g+3 of block db: dv = 32 db + 8 g + 4 hi + 0..3
      constexpr int rr = decltype(Rc)::v, db = rr >> 2, g = rr & 3, reg = 64 * qb + 16 * db + 4 * g;
      T v4[4];
      v4[0] = DT<T>::from_f32(w4_acc_read<reg>() * inv);
      v4[1] = DT<T>::from_f32(w4_acc_read<reg + 1>() * inv);
      v4[2] = DT<T>::from_f32(w4_acc_read<reg + 2>() * inv);
      v4[3] = DT<T>::from_f32(w4_acc_read<reg + 3>() * inv);
      if (qvalid) *reinterpret_cast<u32x2*>(orow + db * 32 + 8 * g + 4 * hi) = *reinterpret_cast<u32x2*>(v4);
    };
    store4(IC<0>{}); store4(IC<1>{}); store4(IC<2>{}); store4(IC<3>{}); store4(IC<4>{}); store4(IC<5>{}); store4(IC<6>{}); store4(IC<7>{});
    store4(IC<8>{}); store4(IC<9>{}); store4(IC<10>{}); store4(IC<11>{}); store4(IC<12>{}); store4(IC<13>{}); store4(IC<14>{}); store4(IC<15>{});
  };
  store_q(IC<0>{});
  store_q(IC<1>{});
}
