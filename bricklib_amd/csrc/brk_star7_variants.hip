// brk_star7_variants.hip -- earlier kernel generations and experiments of brk_star7.hip: compiled ONLY in `make VARIANTS=1` builds (A/B runs
// with tools/kbench through brk_set_variant, and the variant parity tests); the default libbrk.so does not contain a line of this file.
// It is not a translation unit of its own: brk_star7.hip includes it twice under -DBRK_VARIANTS, once for the kernels (part 1, inside
// the anonymous namespace, so that they share the helpers of the shipped kernel) and once for the launch dispatch (part 2).
#if BRK_VARIANTS_PART == 1

__device__ __forceinline__ void issue_loads(Faces &f, const double *__restrict__ in, size_t step, uint32_t b,
                                            const Nbr &nb, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  const uint32_t n_km = __builtin_amdgcn_readfirstlane(nb.km), n_kp = __builtin_amdgcn_readfirstlane(nb.kp);
  const uint32_t n_jm = __builtin_amdgcn_readfirstlane(nb.jm), n_jp = __builtin_amdgcn_readfirstlane(nb.jp);
  const uint32_t n_im = __builtin_amdgcn_readfirstlane(nb.im), n_ip = __builtin_amdgcn_readfirstlane(nb.ip);
  const gcptr self = sbase(in + (size_t)b * step);
  const gcptr p_km = sbase(in + (size_t)n_km * step), p_kp = sbase(in + (size_t)n_kp * step);
  const gcptr p_jm = sbase(in + (size_t)n_jm * step), p_jp = sbase(in + (size_t)n_jp * step);
  const gcptr p_im = sbase(in + (size_t)n_im * step), p_ip = sbase(in + (size_t)n_ip * step);
#pragma unroll
  for (int k = 0; k < 8; ++k) f.own[k] = self[k * 64 + lane];
  f.km = p_km[7 * 64 + lane];          // plane k=7 of the brick below
  f.kp = p_kp[lane];                   // plane k=0 of the brick above
  f.jm = p_jm[hi * 64 + 7 * 8 + lo];   // lane=(k,i): row j=7 of the -j brick
  f.jp = p_jp[hi * 64 + lo];           // row j=0 of the +j brick
  f.im = p_im[hi * 64 + lo * 8 + 7];   // lane=(k,j): column i=7 of the -i brick
  f.ip = p_ip[hi * 64 + lo * 8];       // column i=0 of the +i brick
}

template <int MPI>
__device__ __forceinline__ void compute_store(const double *tile, const Faces &f, double *dst_, int lane,
                                              const Coeffs &cf) {
  const int hi = lane >> 3, lo = lane & 7;
  gptr dst = (gptr)dst_;
  asm volatile("" : "+s"(dst)); // see sbase()
  // per-lane tile offsets of the four in-plane neighbours in plane 0 (plane k adds k*kPlane, an immediate)
  const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
  const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : 80 + hi;
  const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : 88 + hi;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
    const double im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
    const double c0 = f.own[k];
    const double km = k == 0 ? f.km : f.own[k > 0 ? k - 1 : 0];
    const double kp = k == 7 ? f.kp : f.own[k < 7 ? k + 1 : 7];
    double res;
    res = point7<MPI>(kp, km, jp, jm, ip, im, c0, cf);
    store_result(&dst[k * 64 + lane], res);
  }
}

// ---- variant 2: one brick per wave (earlier generation, -DBRK_VARIANTS builds only) --------------------------------
template <int MPI>
__global__ __launch_bounds__(kWG, 6) void k_s7(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                double *__restrict__ out, size_t step,
                                                const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                                Coeffs cf) {
  __shared__ double lds[kWavesPerWG * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * kWavesPerWG + (threadIdx.x >> 6);
  if (slot >= n) return;
  const int lane = threadIdx.x & 63;
  double *tile = lds + (threadIdx.x >> 6) * kTile;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  Faces f;
  issue_loads(f, in, step, b, load_nbr(adj, nullptr, b), lane);
  stage(tile, f, lane);
  wave_lds_sync();
  compute_store<MPI>(tile, f, out + (size_t)b * step, lane, cf);
}

// ==== round-4 experiments, development builds only: two bricks per wave for the 7-point stencil.  Measured at 512^3, interleaved with
// the default kernel (profiles/r04_a_7pt_two_bricks_per_wave.txt, r04_d_7pt_kmarch_one_tile.txt): two tiles, k- or j-paired lists,
// 12 waves/CU 0.382 ms; one tile, k-march, 24 or 16 waves/CU 0.395-0.401 ms; default (one brick per wave, 24 waves/CU) 0.363-0.369 ms
// on the same boxes.  The pairing that pays for the radius-2 and radius-4 stars (k_star_km) does not pay at radius 1: a 7-point wave
// saves one of its twelve loads and pays with half the independent waves per CU.  Kept for A/B runs (brk_set_variant 7..12).
// ---- two bricks per wave ------------------------------------------------------------------------------------------------------
// A workgroup of NW waves takes 2 NW list entries -- a run A of NW entries and the run B behind it -- and wave w handles entry w of
// both: everything the two bricks need is requested up front (twice the bytes in flight per wave), both are staged in their own
// tiles, ONE barrier, then both are computed.  When B is a face neighbour of A the face between them is never requested:
//   B = A's +k neighbour (k-paired lists)   A's upper k-plane IS B.own[0], B's lower k-plane IS A.own[7]: registers of the same lane;
//   B = A's +j neighbour (j-paired lists)   A's +j halo row IS row 0 of B's tile, B's -j halo row IS row 7 of A's tile: the lanes on
//                                           that edge point their ds_read at the partner's tile (a tile is 768 doubles = a multiple
//                                           of the 64 banks, and halo row 9 shares its banks with row 1: still conflict-free).
// A j-face costs 8 half-used 128-byte lines, a k-face 4 full ones; per brick the requests through the vector L1 go from 56 to 52
// (k-pairs) or 48 (j-pairs).  Any list is computed correctly: unpaired entries load all their faces, entries without an i-sibling in
// their run gather their columns into the fallback slots, padding entries idle.
template <int KIND, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void k_s7_2t(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                        double *__restrict__ out, size_t step, const uint32_t *__restrict__ ids,
                                                        uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ faces) {
  __shared__ double lds[2 * NW * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tileA = lds + w * kTile, *tileB = lds + (NW + w) * kTile; // the A tiles side by side (siblings at -+kTile), then the B tiles
  constexpr int kAB = NW * kTile;                                   // tileB - tileA
  auto entry = [&](uint32_t sl) -> uint32_t { return sl < n ? __builtin_amdgcn_readfirstlane(ids ? ids[sl] : first + sl) : BRK_LIST_SKIP; };
  const uint32_t slotA = wg * 2 * NW + w, slotB = slotA + NW;
  const uint32_t bA = entry(slotA), bB = entry(slotB);
  const bool actA = bA != BRK_LIST_SKIP, actB = bB != BRK_LIST_SKIP;
  double ownA[8], ownB[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ownA[k] = ownB[k] = 0.0;
  if (actA) {
    const gcptr self = sbase(in + (size_t)bA * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) ownA[k] = self[k * 64 + lane];
  }
  if (actB) {
    const gcptr self = sbase(in + (size_t)bB * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) ownB[k] = self[k * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0); // own planes are in flight before anything waits for an adjacency row
  Nbr nA{}, nB{};
  if (actA) nA = load_nbr(adj, faces, bA);
  if (actB) nB = load_nbr(adj, faces, bB);
  const uint32_t pA = w > 0 ? entry(slotA - 1) : BRK_LIST_SKIP, qA = w + 1 < NW ? entry(slotA + 1) : BRK_LIST_SKIP;
  const uint32_t pB = w > 0 ? entry(slotB - 1) : BRK_LIST_SKIP, qB = w + 1 < NW ? entry(slotB + 1) : BRK_LIST_SKIP;
  auto u = [](uint32_t x) { return __builtin_amdgcn_readfirstlane(x); };
  const bool shA_im = actA && pA != BRK_LIST_SKIP && u(nA.im) == pA, shA_ip = actA && qA != BRK_LIST_SKIP && u(nA.ip) == qA;
  const bool shB_im = actB && pB != BRK_LIST_SKIP && u(nB.im) == pB, shB_ip = actB && qB != BRK_LIST_SKIP && u(nB.ip) == qB;
  const bool kpaired = actA && actB && u(nA.kp) == bB && u(nB.km) == bA;
  const bool jpaired = actA && actB && u(nA.jp) == bB && u(nB.jm) == bA;
  double kmA = 0.0, kpA = 0.0, kmB = 0.0, kpB = 0.0, jmA = 0.0, jpA = 0.0, jmB = 0.0, jpB = 0.0, imA = 0.0, ipA = 0.0, imB = 0.0, ipB = 0.0;
  if (actA) {
    kmA = sbase(in + (size_t)u(nA.km) * step)[7 * 64 + lane];
    jmA = sbase(in + (size_t)u(nA.jm) * step)[hi * 64 + 7 * 8 + lo];
    if (!kpaired) kpA = sbase(in + (size_t)u(nA.kp) * step)[lane];
    if (!jpaired) jpA = sbase(in + (size_t)u(nA.jp) * step)[hi * 64 + lo];
    if (!shA_im) imA = sbase(in + (size_t)u(nA.im) * step)[hi * 64 + lo * 8 + 7];
    if (!shA_ip) ipA = sbase(in + (size_t)u(nA.ip) * step)[hi * 64 + lo * 8];
  }
  if (actB) {
    kpB = sbase(in + (size_t)u(nB.kp) * step)[lane];
    jpB = sbase(in + (size_t)u(nB.jp) * step)[hi * 64 + lo];
    if (!kpaired) kmB = sbase(in + (size_t)u(nB.km) * step)[7 * 64 + lane];
    if (!jpaired) jmB = sbase(in + (size_t)u(nB.jm) * step)[hi * 64 + 7 * 8 + lo];
    if (!shB_im) imB = sbase(in + (size_t)u(nB.im) * step)[hi * 64 + lo * 8 + 7];
    if (!shB_ip) ipB = sbase(in + (size_t)u(nB.ip) * step)[hi * 64 + lo * 8];
  }
  // ---- stage both bricks
  if (actA) {
#pragma unroll
    for (int k = 0; k < 8; ++k) tileA[k * kPlane + (hi + 1) * 8 + lo] = ownA[k];
    tileA[hi * kPlane + lo] = jmA;
    if (!jpaired) tileA[hi * kPlane + 72 + lo] = jpA;
    if (!shA_im) tileA[hi * kPlane + 80 + lo] = imA;
    if (!shA_ip) tileA[hi * kPlane + 88 + lo] = ipA;
  }
  if (actB) {
#pragma unroll
    for (int k = 0; k < 8; ++k) tileB[k * kPlane + (hi + 1) * 8 + lo] = ownB[k];
    tileB[hi * kPlane + 72 + lo] = jpB;
    if (!jpaired) tileB[hi * kPlane + lo] = jmB;
    if (!shB_im) tileB[hi * kPlane + 80 + lo] = imB;
    if (!shB_ip) tileB[hi * kPlane + 88 + lo] = ipB;
  }
  __syncthreads(); // every tile of the group is complete
  if (actA) {
    gptr dst = (gptr)(out + (size_t)bA * step);
    asm volatile("" : "+s"(dst));
    const int o_jm = hi * 8 + lo;
    const int o_jp = (hi < 7 || !jpaired) ? (hi + 2) * 8 + lo : kAB + 8 + lo; // row 0 of the partner's brick = its tile row 1
    const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (shA_im ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
    const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (shA_ip ? kTile + (hi + 1) * 8 : 88 + hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double jm = tileA[k * kPlane + o_jm], jp = tileA[k * kPlane + o_jp];
      const double im = tileA[k * kPlane + o_im], ip = tileA[k * kPlane + o_ip];
      const double km = k == 0 ? kmA : ownA[k > 0 ? k - 1 : 0];
      const double kp = k == 7 ? (kpaired ? ownB[0] : kpA) : ownA[k < 7 ? k + 1 : 7];
      store_result(&dst[k * 64 + lane], point7<KIND>(kp, km, jp, jm, ip, im, ownA[k], cf));
    }
  }
  if (actB) {
    gptr dst = (gptr)(out + (size_t)bB * step);
    asm volatile("" : "+s"(dst));
    const int o_jm = (hi > 0 || !jpaired) ? hi * 8 + lo : -kAB + 8 * 8 + lo; // row 7 of the partner's brick = its tile row 8
    const int o_jp = (hi + 2) * 8 + lo;
    const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (shB_im ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
    const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (shB_ip ? kTile + (hi + 1) * 8 : 88 + hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double jm = tileB[k * kPlane + o_jm], jp = tileB[k * kPlane + o_jp];
      const double im = tileB[k * kPlane + o_im], ip = tileB[k * kPlane + o_ip];
      const double km = k == 0 ? (kpaired ? ownA[7] : kmB) : ownB[k > 0 ? k - 1 : 0];
      const double kp = k == 7 ? kpB : ownB[k < 7 ? k + 1 : 7];
      store_result(&dst[k * 64 + lane], point7<KIND>(kp, km, jp, jm, ip, im, ownB[k], cf));
    }
  }
}


// ---- two k-consecutive bricks per wave, one tile (the 7-point form of k_star_km, brk_starwg.hip) -------------------------------
// A workgroup of NW waves takes 2 NW list entries: a run of NW i-neighbours and the run of their +k neighbours (k-paired lists with
// runs of NW).  Wave w requests both bricks and everything brick 0 needs up front, requests the faces of brick 1 just before brick 0
// is staged, computes brick 0, then brick 1 out of the same tile.  The k-face between the two never travels (11 instead of 12 loads
// per brick), and brick 1's requests are in flight through brick 0's staging, barrier and arithmetic.  Groups that are not two full
// k-linked runs (the workgroup votes) take one brick at a time.
template <int KIND, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void k_s7_km(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                        double *__restrict__ out, size_t step, const uint32_t *__restrict__ ids,
                                                        uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ faces) {
  __shared__ double lds[NW * kTile];
  __shared__ uint32_t vote[NW];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * kTile;
  auto entry = [&](uint32_t sl) -> uint32_t { return sl < n ? __builtin_amdgcn_readfirstlane(ids ? ids[sl] : first + sl) : BRK_LIST_SKIP; };
  auto u = [](uint32_t x) { return __builtin_amdgcn_readfirstlane(x); };
  const uint32_t slot0 = wg * 2 * NW + w, slot1 = slot0 + NW;
  const uint32_t b0 = entry(slot0), b1 = entry(slot1);
  bool chain = b0 != BRK_LIST_SKIP && b1 != BRK_LIST_SKIP;
  double own0[8], own1[8];
  {
    const gcptr s0 = sbase(in + (size_t)(chain ? b0 : 0u) * step), s1 = sbase(in + (size_t)(chain ? b1 : 0u) * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) own0[k] = s0[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) own1[k] = s1[k * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  const Nbr n0 = load_nbr(adj, faces, chain ? b0 : 0u), n1 = load_nbr(adj, faces, chain ? b1 : 0u);
  chain = chain && u(n0.kp) == b1 && u(n1.km) == b0;
  if (lane == 0) vote[w] = chain ? 1u : 0u;
  __syncthreads();
  bool fast = true;
#pragma unroll
  for (int q = 0; q < NW; ++q) fast = fast && vote[q] != 0u;

  auto compute = [&](uint32_t b, const double (&own)[8], double km, double kp, bool s_im, bool s_ip) {
    gptr dst = (gptr)(out + (size_t)b * step);
    asm volatile("" : "+s"(dst));
    const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
    const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (s_im ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
    const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (s_ip ? kTile + (hi + 1) * 8 : 88 + hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
      const double im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
      const double below = k == 0 ? km : own[k > 0 ? k - 1 : 0], above = k == 7 ? kp : own[k < 7 ? k + 1 : 7];
      store_result(&dst[k * 64 + lane], point7<KIND>(above, below, jp, jm, ip, im, own[k], cf));
    }
  };
  auto stage = [&](const double (&own)[8], double jm, double jp, double im, double ip, bool s_im, bool s_ip) {
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k * kPlane + (hi + 1) * 8 + lo] = own[k];
    tile[hi * kPlane + lo] = jm;
    tile[hi * kPlane + 72 + lo] = jp;
    if (!s_im) tile[hi * kPlane + 80 + lo] = im;
    if (!s_ip) tile[hi * kPlane + 88 + lo] = ip;
  };
  if (fast) {
    const uint32_t p0 = w > 0 ? entry(slot0 - 1) : BRK_LIST_SKIP, q0 = w + 1 < NW ? entry(slot0 + 1) : BRK_LIST_SKIP;
    const uint32_t p1 = w > 0 ? entry(slot1 - 1) : BRK_LIST_SKIP, q1 = w + 1 < NW ? entry(slot1 + 1) : BRK_LIST_SKIP;
    const bool s0m = p0 != BRK_LIST_SKIP && u(n0.im) == p0, s0p = q0 != BRK_LIST_SKIP && u(n0.ip) == q0;
    const bool s1m = p1 != BRK_LIST_SKIP && u(n1.im) == p1, s1p = q1 != BRK_LIST_SKIP && u(n1.ip) == q1;
    const double km0 = sbase(in + (size_t)u(n0.km) * step)[7 * 64 + lane];
    const double jm0 = sbase(in + (size_t)u(n0.jm) * step)[hi * 64 + 7 * 8 + lo], jp0 = sbase(in + (size_t)u(n0.jp) * step)[hi * 64 + lo];
    double im0 = 0.0, ip0 = 0.0, im1 = 0.0, ip1 = 0.0;
    if (!s0m) im0 = sbase(in + (size_t)u(n0.im) * step)[hi * 64 + lo * 8 + 7];
    if (!s0p) ip0 = sbase(in + (size_t)u(n0.ip) * step)[hi * 64 + lo * 8];
    // brick 1's requests: in flight while brick 0 is staged, waited for at the barrier, and computed
    const double kp1 = sbase(in + (size_t)u(n1.kp) * step)[lane];
    const double jm1 = sbase(in + (size_t)u(n1.jm) * step)[hi * 64 + 7 * 8 + lo], jp1 = sbase(in + (size_t)u(n1.jp) * step)[hi * 64 + lo];
    if (!s1m) im1 = sbase(in + (size_t)u(n1.im) * step)[hi * 64 + lo * 8 + 7];
    if (!s1p) ip1 = sbase(in + (size_t)u(n1.ip) * step)[hi * 64 + lo * 8];
    stage(own0, jm0, jp0, im0, ip0, s0m, s0p);
    __syncthreads();
    compute(b0, own0, km0, own1[0], s0m, s0p);
    __syncthreads();
    stage(own1, jm1, jp1, im1, ip1, s1m, s1p);
    __syncthreads();
    compute(b1, own1, own0[7], kp1, s1m, s1p);
    return;
  }
  // ---- slow path: one brick at a time with private k-faces (any list)
#pragma unroll 1
  for (int q = 0; q < 2; ++q) {
    const uint32_t slot = q ? slot1 : slot0;
    const uint32_t b = entry(slot);
    const bool active = b != BRK_LIST_SKIP;
    double own[8], km = 0.0, kp = 0.0;
    bool s_im = false, s_ip = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) own[k] = 0.0;
    if (active) {
      const Nbr nb = load_nbr(adj, faces, b);
      const uint32_t prev = w > 0 ? entry(slot - 1) : BRK_LIST_SKIP, next = w + 1 < NW ? entry(slot + 1) : BRK_LIST_SKIP;
      s_im = prev != BRK_LIST_SKIP && u(nb.im) == prev;
      s_ip = next != BRK_LIST_SKIP && u(nb.ip) == next;
      const gcptr self = sbase(in + (size_t)b * step);
#pragma unroll
      for (int k = 0; k < 8; ++k) own[k] = self[k * 64 + lane];
      km = sbase(in + (size_t)u(nb.km) * step)[7 * 64 + lane];
      kp = sbase(in + (size_t)u(nb.kp) * step)[lane];
      const double jm = sbase(in + (size_t)u(nb.jm) * step)[hi * 64 + 7 * 8 + lo], jp = sbase(in + (size_t)u(nb.jp) * step)[hi * 64 + lo];
      double im = 0.0, ip = 0.0;
      if (!s_im) im = sbase(in + (size_t)u(nb.im) * step)[hi * 64 + lo * 8 + 7];
      if (!s_ip) ip = sbase(in + (size_t)u(nb.ip) * step)[hi * 64 + lo * 8];
      stage(own, jm, jp, im, ip, s_im, s_ip);
    }
    __syncthreads();
    if (active) compute(b, own, km, kp, s_im, s_ip);
    __syncthreads();
  }
}


#elif BRK_VARIANTS_PART == 2
// inside launch_star7: sets `handled` when `variant` names one of the kernels above (or the default kernel at the other group size)
  handled = true;
  if (variant == 10 || variant == 11 || variant == 12) { // two k-consecutive bricks per wave, one tile: 10 = 8-wave groups at 24 waves/CU, 12 = at 16 waves/CU, 11 = 4-wave groups
#define BRK_LAUNCH_KM(K)                                                                                                                             \
  do {                                                                                                                                               \
    if (variant == 10) hipLaunchKernelGGL((k_s7_km<K, 8, 6>), dim3((((n + 15) / 16 + 7) / 8) * 8), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
    else if (variant == 12) hipLaunchKernelGGL((k_s7_km<K, 8, 4>), dim3((((n + 15) / 16 + 7) / 8) * 8), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
    else hipLaunchKernelGGL((k_s7_km<K, 4, 6>), dim3((((n + 7) / 8 + 7) / 8) * 8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
  } while (0)
    if (kind == kMpi) BRK_LAUNCH_KM(kMpi);
    else if (kind == kCond) BRK_LAUNCH_KM(kCond);
    else BRK_LAUNCH_KM(kCoeff);
#undef BRK_LAUNCH_KM
  } else if (variant == 7 || variant == 8 || variant == 9) { // two bricks per wave (two tiles): 7 = 4-wave groups at 3 per CU, 8 = 4-wave groups at 2 per CU, 9 = 8-wave groups
#define BRK_LAUNCH_2T(K)                                                                                                                              \
  do {                                                                                                                                                \
    if (variant == 7) hipLaunchKernelGGL((k_s7_2t<K, 4, 3>), dim3((((n + 7) / 8 + 7) / 8) * 8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
    else if (variant == 8) hipLaunchKernelGGL((k_s7_2t<K, 4, 2>), dim3((((n + 7) / 8 + 7) / 8) * 8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
    else hipLaunchKernelGGL((k_s7_2t<K, 8, 1>), dim3((((n + 15) / 16 + 7) / 8) * 8), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces); \
  } while (0)
    if (kind == kMpi) BRK_LAUNCH_2T(kMpi);
    else if (kind == kCond) BRK_LAUNCH_2T(kCond);
    else BRK_LAUNCH_2T(kCoeff);
#undef BRK_LAUNCH_2T
  } else if (variant == 13) { // round 5: the default kernel with the +-i neighbours by DPP (k_s7_wg<..., IDPP = 1>), list / range form
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
#define BRK_LAUNCH_DPP(K)                                                                                                                                     \
  do {                                                                                                                                                        \
    if (ids) hipLaunchKernelGGL((k_s7_wg<K, 8, kModeList, 0, 1>), dim3(wgs), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, Box{}, cf, nullptr, faces); \
    else hipLaunchKernelGGL((k_s7_wg<K, 8, kModeRange, 0, 1>), dim3(wgs), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, Box{}, cf, nullptr, faces);   \
  } while (0)
    brki_note_kernel("k_s7_wg<.,8,.,0,IDPP=1>");
    if (kind == kMpi) BRK_LAUNCH_DPP(kMpi);
    else if (kind == kCond) BRK_LAUNCH_DPP(kCond);
    else BRK_LAUNCH_DPP(kCoeff);
#undef BRK_LAUNCH_DPP
  } else if (variant == 6) {
    const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
    BRK_LAUNCH_WG_W(4);
  } else if (variant == 2) {
    const uint32_t wgs = (((n + kWavesPerWG - 1) / kWavesPerWG + 7) / 8) * 8;
    if (kind == kMpi)
      hipLaunchKernelGGL(k_s7<kMpi>, dim3(wgs), dim3(kWG), g_pad_lds, st, adj, in, out, step, ids, first, n, cf);
    else if (kind == kCond)
      hipLaunchKernelGGL(k_s7<kCond>, dim3(wgs), dim3(kWG), g_pad_lds, st, adj, in, out, step, ids, first, n, cf);
    else
      hipLaunchKernelGGL(k_s7<kCoeff>, dim3(wgs), dim3(kWG), g_pad_lds, st, adj, in, out, step, ids, first, n, cf);
  }
  else handled = false;
#endif
