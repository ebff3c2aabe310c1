// brk_starwg_variants.hip -- earlier kernel generations and experiments of brk_starwg.hip: compiled ONLY in `make VARIANTS=1` builds (A/B runs
// with tools/kbench through brk_set_variant, and the variant parity tests); the default libbrk.so does not contain a line of this file.
// It is not a translation unit of its own: brk_starwg.hip includes it twice under -DBRK_VARIANTS, once for the kernels (part 1, inside
// the anonymous namespace, so that they share the helpers of the shipped kernel) and once for the launch dispatch (part 2).
#if BRK_VARIANTS_PART == 1

// ---- two bricks per wave: k-pairs (round 3's kernel; superseded by k_star_km<NB = 2> below, kept for A/B runs) --------------------
// k_star_wg above is bound by what fits on a CU: a radius-4 wave holds 12 KiB of LDS, so 12 waves = 12 bricks are in flight per CU,
// and each of them requests its 2R k-halo planes although half of those are the OWN planes of the brick above or below.  Here a
// workgroup of 4 waves takes EIGHT list entries -- a run of four i-neighbours (A) and the run behind it in the list (B) -- and a wave
// handles entry w of each run.  Everything both bricks need is requested up front (twice the bytes in flight per wave and per KiB
// of LDS: the tile is used for A, then reused for B), and when B is the +k neighbour of A (lists in the k-paired order of
// bricklib_amd/order.py / the box entry point) the planes between them are loaded once: A's upper halo IS B's own planes 0..R-1
// and B's lower halo IS A's planes 8-R..7, all in the same register column P[].  Per pair 16 + 2R planes instead of 16 + 4R.
// Any list is computed correctly: unpaired entries load their own halos (aboveA / belowB), entries without an i-sibling gather
// their columns into the fallback slots, inactive entries (padding, list end) skip their phase.
template <int KIND, int OCC>
__global__ __launch_bounds__(256, OCC) void k_star_kp(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ faces) {
  constexpr int R = StarTraits<KIND>::R, H = R / 2, NW = 4;
  using TL = WgTile<R>;
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) d2 *gc2;
  __shared__ double lds[NW * TL::TILE];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * TL::TILE;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };
  auto entry = [&](uint32_t sl) -> uint32_t { return sl < n ? __builtin_amdgcn_readfirstlane(id_of(sl)) : BRK_LIST_SKIP; };

  struct Run { // one of the two bricks of this wave
    uint32_t b, km, kp, jm, jp, im, ip;
    bool active, share_im, share_ip;
  };
  auto open_run = [&](uint32_t slot) {
    Run r{};
    r.b = entry(slot);
    r.active = r.b != BRK_LIST_SKIP;
    return r;
  };
  const uint32_t slotA = wg * 8 + w, slotB = slotA + 4;
  Run A = open_run(slotA), B = open_run(slotB);
  double P[16 + 2 * R];              // planes: [0,R) below A | [R,R+8) A | [R+8,R+16) B | [R+16,R+16+R) above B
  double aboveA[R], belowB[R];       // only for unpaired entries
#pragma unroll
  for (int q = 0; q < 16 + 2 * R; ++q) P[q] = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) aboveA[r] = belowB[r] = 0.0;
  // own planes need only the ids: in flight before any adjacency is waited for
  if (A.active) {
    const gcptr self = sbase(in + (size_t)A.b * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) P[R + k] = self[k * 64 + lane];
  }
  if (B.active) {
    const gcptr self = sbase(in + (size_t)B.b * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) P[R + 8 + k] = self[k * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  auto neighbours = [&](Run &r, uint32_t slot) {
    if (!r.active) return;
    if (faces) { // two branches, each through one restrict pointer: scalar loads (see k_star_wg)
      const uint32_t *f = faces + (size_t)r.b * 8;
      r.km = __builtin_amdgcn_readfirstlane(f[0]), r.kp = __builtin_amdgcn_readfirstlane(f[1]);
      r.jm = __builtin_amdgcn_readfirstlane(f[2]), r.jp = __builtin_amdgcn_readfirstlane(f[3]);
      r.im = __builtin_amdgcn_readfirstlane(f[4]), r.ip = __builtin_amdgcn_readfirstlane(f[5]);
    } else {
      const uint32_t *row = adj + (size_t)r.b * 27;
      r.km = __builtin_amdgcn_readfirstlane(row[adj_slot(-1, 0, 0)]), r.kp = __builtin_amdgcn_readfirstlane(row[adj_slot(1, 0, 0)]);
      r.jm = __builtin_amdgcn_readfirstlane(row[adj_slot(0, -1, 0)]), r.jp = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 1, 0)]);
      r.im = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, -1)]), r.ip = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, 1)]);
    }
    const uint32_t prev = w > 0 ? entry(slot - 1) : BRK_LIST_SKIP, next = w + 1 < NW ? entry(slot + 1) : BRK_LIST_SKIP;
    r.share_im = prev != BRK_LIST_SKIP && r.im == prev;
    r.share_ip = next != BRK_LIST_SKIP && r.ip == next;
  };
  neighbours(A, slotA);
  neighbours(B, slotB);
  const bool paired = A.active && B.active && A.kp == B.b && B.km == A.b;
  // ---- k-halos
  if (A.active) {
    const gcptr p = sbase(in + (size_t)A.km * step);
#pragma unroll
    for (int r = 0; r < R; ++r) P[r] = p[(8 - R + r) * 64 + lane];
  }
  if (B.active) {
    const gcptr p = sbase(in + (size_t)B.kp * step);
#pragma unroll
    for (int r = 0; r < R; ++r) P[R + 16 + r] = p[r * 64 + lane];
  }
  if (!paired) {
    if (A.active) {
      const gcptr p = sbase(in + (size_t)A.kp * step);
#pragma unroll
      for (int r = 0; r < R; ++r) aboveA[r] = p[r * 64 + lane];
    }
    if (B.active) {
      const gcptr p = sbase(in + (size_t)B.km * step);
#pragma unroll
      for (int r = 0; r < R; ++r) belowB[r] = p[(8 - R + r) * 64 + lane];
    }
  }
  // ---- j- and i-faces of both bricks, 16-byte lanes (every 128-byte line requested by one instruction, see k_star_wg)
  const int ik = R == 2 ? hi : (lane >> 4), ij = R == 2 ? lo : ((lane >> 1) & 7), ih = R == 2 ? 0 : (lane & 1);
  struct FaceRegs {
    d2 jm[H], jp[H], im[H], ip[H];
  };
  auto load_faces = [&](const Run &r, FaceRegs &f) {
#pragma unroll
    for (int q = 0; q < H; ++q) f.jm[q] = f.jp[q] = f.im[q] = f.ip[q] = d2{0.0, 0.0};
    if (!r.active) return;
    const gcptr p_jm = sbase(in + (size_t)r.jm * step), p_jp = sbase(in + (size_t)r.jp * step);
#pragma unroll
    for (int q = 0; q < H; ++q) {
      f.jm[q] = *(gc2)(p_jm + hi * 64 + (8 - R) * 8 + q * 16 + 2 * lo);
      f.jp[q] = *(gc2)(p_jp + hi * 64 + q * 16 + 2 * lo);
    }
    if (!r.share_im) {
      const gcptr p_im = sbase(in + (size_t)r.im * step);
#pragma unroll
      for (int q = 0; q < H; ++q) f.im[q] = *(gc2)(p_im + (ik + 4 * q) * 64 + ij * 8 + (8 - R) + 2 * ih);
    }
    if (!r.share_ip) {
      const gcptr p_ip = sbase(in + (size_t)r.ip * step);
#pragma unroll
      for (int q = 0; q < H; ++q) f.ip[q] = *(gc2)(p_ip + (ik + 4 * q) * 64 + ij * 8 + 2 * ih);
    }
  };
  FaceRegs fA, fB;
  load_faces(A, fA);
  load_faces(B, fB);

  // ---- one phase: stage the brick whose planes start at P[OFF + R], meet the siblings, compute, store
  auto stage = [&](const Run &r, const FaceRegs &f, auto own) {
    if (!r.active) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k * TL::PLANE + (hi + R) * 8 + lo] = own(k);
#pragma unroll
    for (int q = 0; q < H; ++q) {
      *(d2 *)&tile[hi * TL::PLANE + (2 * q) * 8 + 2 * lo] = f.jm[q];
      *(d2 *)&tile[hi * TL::PLANE + (R + 8 + 2 * q) * 8 + 2 * lo] = f.jp[q];
    }
    if (!r.share_im) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih) * 8 + ij] = f.im[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih + 1) * 8 + ij] = f.im[q].y;
      }
    }
    if (!r.share_ip) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih) * 8 + ij] = f.ip[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih + 1) * 8 + ij] = f.ip[q].y;
      }
    }
  };
  auto compute = [&](const Run &r, auto colv) { // colv(idx): the lane's k-column, idx in [0, 8 + 2R), own planes at [R, R + 8)
    if (!r.active) return;
    gptr dst = (gptr)(out + (size_t)r.b * step);
    asm volatile("" : "+s"(dst));
    int o_im[R], o_ip[R];
#pragma unroll
    for (int q = 1; q <= R; ++q) {
      const int cm = lo - q, cp = lo + q;
      o_im[q - 1] = cm >= 0 ? (hi + R) * 8 + cm : (r.share_im ? -TL::TILE + (hi + R) * 8 + cm + 8 : TL::IM + (cm + R) * 8 + hi);
      o_ip[q - 1] = cp < 8 ? (hi + R) * 8 + cp : (r.share_ip ? TL::TILE + (hi + R) * 8 + cp - 8 : TL::IP + (cp - 8) * 8 + hi);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double jm[R], jp[R], im[R], ip[R];
#pragma unroll
      for (int q = 1; q <= R; ++q) {
        jm[q - 1] = tile[k * TL::PLANE + (hi + R - q) * 8 + lo];
        jp[q - 1] = tile[k * TL::PLANE + (hi + R + q) * 8 + lo];
        im[q - 1] = tile[k * TL::PLANE + o_im[q - 1]];
        ip[q - 1] = tile[k * TL::PLANE + o_ip[q - 1]];
      }
      double res = 0.0; // stencils/fake.h:41-53 (25-pt) / :73-79 (13-pt): outermost shell first, centre last
#pragma unroll
      for (int q = R; q >= 1; --q) {
        const double sh = colv(R + k + q) + colv(R + k - q) + jp[q - 1] + jm[q - 1] + ip[q - 1] + im[q - 1];
        res = (q == R) ? sh * cf.c[q] : res + sh * cf.c[q];
      }
      res = res + colv(R + k) * cf.c[0];
      store_result(&dst[k * 64 + lane], res);
    }
  };
  stage(A, fA, [&](int k) __attribute__((always_inline)) { return P[R + k]; });
  __syncthreads(); // A's tiles are complete
  compute(A, [&](int idx) __attribute__((always_inline)) { return (idx < R + 8 || paired) ? P[idx] : aboveA[idx - R - 8 < 0 ? 0 : idx - R - 8]; });
  __syncthreads(); // every wave has finished reading the A tiles
  stage(B, fB, [&](int k) __attribute__((always_inline)) { return P[R + 8 + k]; });
  __syncthreads(); // B's tiles are complete
  compute(B, [&](int idx) __attribute__((always_inline)) { return (idx >= R || paired) ? P[8 + idx] : belowB[idx < R ? idx : 0]; });
}


#elif BRK_VARIANTS_PART == 2
// inside launch_star_wg: early returns for the variants that name one of the kernels above or another group size / occupancy
  if (stencil_id == BRK_MPI13 && variant == 2) {
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_star_wg<kStarMpi13, 8, 4>), dim3(wgs), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  if (stencil_id == BRK_MPI25 && variant == 2) {
    const uint32_t wgs = (((n + 1) / 2 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_star_wg<kStarMpi25, 2, 3>), dim3(wgs), dim3(128), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  if (variant == 4 || variant == 5) { // experiments: 4 = four k-consecutive bricks per wave (k-stacked lists), 5 = the 13-point march at 5 waves per SIMD
    const uint32_t per = variant == 4 ? 16 : 8;
    const uint32_t wgsm = (((n + per - 1) / per + 7) / 8) * 8;
    if (stencil_id == BRK_MPI13 && variant == 4) hipLaunchKernelGGL((k_star_km<kStarMpi13, 4, 4>), dim3(wgsm), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    else if (stencil_id == BRK_MPI13) hipLaunchKernelGGL((k_star_km<kStarMpi13, 2, 5>), dim3(wgsm), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    else if (variant == 4) hipLaunchKernelGGL((k_star_km<kStarMpi25, 4, 3>), dim3(wgsm), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    else hipLaunchKernelGGL((k_star_km<kStarMpi25, 2, 4>), dim3(wgsm), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  if (variant == 3) {
    const uint32_t wgs8 = (((n + 7) / 8 + 7) / 8) * 8;
    if (stencil_id == BRK_MPI13) hipLaunchKernelGGL((k_star_kp<kStarMpi13, 4>), dim3(wgs8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    else hipLaunchKernelGGL((k_star_kp<kStarMpi25, 3>), dim3(wgs8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
#endif
