// s2m_register.hpp — k_register, the registration kernel: one launch = one surfOptimization() pass
// (reference src/mapOptmization.cpp:1074-1143) fused with the matA/matB row assembly of LMOptimization()
// (:1191-1235) and the first stage of the AtA / AtB reduction (:1237-1239); in the fused loop its
// prologue also closes the previous LM iteration (lm_close_iteration).  Included by s2m_kernels.hpp.
//
// What the reference recomputes 30 times per scan, and what actually changes
// ---------------------------------------------------------------------------
// The LM loop revisits the same scan with an ever smaller pose step.  surfOptimization() redoes the
// 5-NN search, the plane fit and the inlier test for every point in every iteration, although for
// nearly every point the ORDERED neighbour tuple - and with it the plane, which depends on nothing
// else - is the same as in the iteration before.  This kernel keeps, per scan point, what it needs to
// PROVE that, and touches the map only for the points where the proof fails:
//
//   tier A  certificate.  When a point's tuple (n1..n5, ascending (d2, map index)) is established at
//           the transformed position q_ref, the kernel also knows r6: no other map point was nearer
//           than r6.  With d1..d5 the tuple's distances, let
//               slack = min( (d2-d1)/2, .., (d5-d4)/2, (r6-d5)/2, gate-d5 ) - margin.
//           At a later pose the point stands at q with e = |q - q_ref|.  Every distance changed by at
//           most e (triangle inequality), so while e < slack the order of n1..n5 is the same, every
//           other map point is still farther than n5, and d5 is still inside the gate (:1097): same
//           tuple, same plane, same inlier verdict - bit for bit what a fresh search would return,
//           without reading a single map point.  A point with fewer than 5 map points inside the
//           gate carries the mirror image: slack = r6 - gate - margin, "still not gated".
//           The margin (2 um) is twenty times the fp32 rounding of the distance arithmetic (1e-7 relative
//           on distances below 1 m; the triangle inequality itself is exact for the rounded positions the
//           reference measures from); tuples with an exact tie (gap 0) never get a certificate.
//   tier B  re-measure.  Besides its tuple a point keeps its FRONT: the (up to) six nearest map points it knew when it last
//           searched - EVERY map point within r_out of q_ref - with their coordinates (96 bytes).  A point that moved farther
//           than its slack measures those six again, without touching the map: everything else is at least r_out - e away, so
//           if the new 5th distance is below that the tuple is proved; order, gate and slack are refreshed, the plane is
//           refitted only if the order changed.  A point the front does not settle takes the 5th distance along as the bound
//           of its search.
//   tier C  search.  Only the lanes that fail A and B search.  The box rows the lanes of a pass need are streamed through a
//           filter into the wave's LDS tile (sixteen rows in flight per round trip); every lane then keeps the SEVEN nearest
//           tile points in a sorted list of 32-bit keys - the bits of the squared distance with the low 9 replaced by the tile
//           slot - by one v_med3_u32 per list element and tile point (near7_insert): 17 instructions per lane and tile point,
//           whatever the lane finds (round 3's count / list / select sweeps: ~80).  The six nearest by key are then measured
//           exactly from the tile (LDS) and put in exact order; the 7th key's distance part bounds everything else from below,
//           and the answer stands if the exact 5th distance is below that bound (a near-tie the keys cannot resolve - 18 um at
//           0.6 m - is served exactly instead).  Idle lanes share the sweep of passes with few searching lanes; a box that
//           overflows the tile is swept in chunks with a running list (256-register builds) or split into passes of fewer
//           lanes (128-register builds); a handful of leftover lanes are served one by one (the wave reads a lane's <= 9
//           x-runs cooperatively); the lanes of a wave too scattered to share anything - a sparse far-field ring segment -
//           walk their own cells, every lane for itself and all at once.  Exact keys are (fp32 d2 bits << 32 | map position);
//           the reference order (d2, original index) differs only between points at exactly equal distance: ties inside the
//           six are re-ordered after a fetch from the map, a tie across the 5th/6th boundary is settled by an exact walk.
//
// In the steady state of the loop (pose steps of 1e-4 m and less) nearly every wave is all tier A: it
// reads 44 bytes per point (position, certificate, plane) and no map data at all.
// Round 4 rebuilt tiers B and C (fronts with coordinates, the key sweep): launch 0 of a kitti64 scan 71 -> 40 us.
//
// Two passes per wave, so that the 28 fp64 sums of the normal equations never coexist with the search's
// registers:  pass 1 (associate) brings every point's tuple / plane / certificate up to date for this
// pose;  pass 2 (linearise) streams position + plane: residual, weight (:1125-1139), Jacobian row
// (:1216-1234), 21+6+1 products accumulated in fp64.  A wave loops over wave-table entries (stride =
// waves in the grid), so the grid never exceeds kMaxBlocks co-resident workgroups whatever the scan
// size, and the accumulators are reduced once per wave: recursive halving over the lanes, LDS across
// the 8 waves, one partial row per workgroup in the slot of this launch's parity.
// combineOptimizationCoeffs() (:1145-1156) has no counterpart: rejected lanes add zeros.
#pragma once
// (included inside namespace s2m by s2m_kernels.hpp)

constexpr float kCertMargin = 2e-6f;   // metres; see tier A above
constexpr float kNbrReach   = 0.15f;   // metres beyond the 5th neighbour that a search looks, so that what it learns about "everything else" reaches that far ...
constexpr float kTightBound = 0.62f;   // metres: a front whose 5th member is nearer than this now is taken to be (nearly) the answer
constexpr float kNbrReachCold = 0.10f; // ... and beyond the gate when it has nothing to start from (first launch of a scan)
constexpr int   kWalkLanes = 64;       // a wave serves up to this many leftover lanes one by one; beyond, every lane walks its own cells
constexpr int   kNbr = kNbrCap;        // members of a point's front: the six nearest map points it knew
constexpr float kFragileSlack = 3e-5f; // a certificate with less slack than this may well fail in the steady state of the loop: such a lane prefetches
constexpr float kSlabRound = 1e-6f;    // rounding of the cell binning per metre of distance from the grid origin (3 roundings of 6e-8 each, see kSlabMargin)
constexpr float kAbsRound = 2.4e-7f;   // two ulps of an absolute coordinate, per metre of it
constexpr int   kTilePad = 32;         // far-away entries behind a tile's last point (four steps of up to eight parts)
constexpr int   kShareMin = 24;        // tile points from which idle lanes share the sweep of a pass with few searching lanes
constexpr int   kWalkMax = 384;        // a scattered wave's lanes walk their own cells (in teams: idle lanes join in) if none has more candidates than this
constexpr int   kServeLanes = 4;       // up to this many leftover lanes of a pass that cannot be staged are served one by one; more are split further
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));

__device__ __forceinline__ uint32_t key_hi(uint64_t k) { return (uint32_t)(k >> 32); }
__device__ __forceinline__ uint32_t key_lo(uint64_t k) { return (uint32_t)k; }
__device__ __forceinline__ void cas_u64(uint64_t& a, uint64_t& b)
{
    const bool c = b < a;
    const uint64_t t = a;
    a = c ? b : a; b = c ? t : b;
}

// the 6 smallest keys seen, ascending: key = (fp32 d2 bits << 32) | position in map_sorted
struct Top6k { uint64_t key[6]; };
// returns the key that stays outside: the smallest of those over all insertions is the 7th nearest
__device__ __forceinline__ uint64_t top6k_insert(Top6k& t, uint64_t key)
{
    cas_u64(t.key[5], key);            // key > t.key[5]: nothing changes below
#pragma unroll
    for (int j = 5; j > 0; --j) cas_u64(t.key[j - 1], t.key[j]);
    return key;
}

// six keys in any order -> ascending: the 12-comparator network for six inputs (a third of six insertions)
__device__ __forceinline__ void sort6k(Top6k& t)
{
    cas_u64(t.key[0], t.key[5]); cas_u64(t.key[1], t.key[3]); cas_u64(t.key[2], t.key[4]);
    cas_u64(t.key[1], t.key[2]); cas_u64(t.key[3], t.key[4]);
    cas_u64(t.key[0], t.key[3]); cas_u64(t.key[2], t.key[5]);
    cas_u64(t.key[0], t.key[1]); cas_u64(t.key[2], t.key[3]); cas_u64(t.key[4], t.key[5]);
    cas_u64(t.key[1], t.key[2]); cas_u64(t.key[3], t.key[4]);
}

// exact top-5 of the fallback sweep (a neighbourhood that would not fit): same keys
struct Top5k { uint64_t key[5]; };
__device__ __forceinline__ void top5k_insert(Top5k& t, uint64_t key)
{
    t.key[4] = key;
#pragma unroll
    for (int j = 4; j > 0; --j) cas_u64(t.key[j - 1], t.key[j]);
}

// m = {x, y, z, low word of the key}: a tile entry carries the map position there, a map_sorted entry the original index
__device__ __forceinline__ uint64_t make_key(const v4f m, float sx, float sy, float sz, float& d2)
{
    const float dx = sx - m.x, dy = sy - m.y, dz = sz - m.z;
    d2 = (dx * dx + dy * dy) + dz * dz;                                               // L2_Simple order
    return ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)__float_as_int(m.w);
}

// bound = min(d2 of the current 5th best, gate): nothing at or beyond the gate is observable.
// tieb remembers the d2 bits of a point that was met at exactly the 5th best's distance (or pushed out of the set at the
// new 5th distance): if that is still the 5th distance at the end, (d2, position) and (d2, original index) may disagree
// about who is fifth.
__device__ __forceinline__ void consider(Top5k& best, float& bound, float gatef, uint32_t& tieb, const v4f m, float sx, float sy, float sz)
{
    float d2;
    const uint64_t key = make_key(m, sx, sy, sz, d2);
    const uint32_t pos = key_lo(key), d2b = key_hi(key);
    const bool fresh = (pos != key_lo(best.key[0])) & (pos != key_lo(best.key[1])) & (pos != key_lo(best.key[2])) & (pos != key_lo(best.key[3]));
    const bool inb = d2 <= bound;
    const bool tie = inb & fresh & (d2b == key_hi(best.key[4])) & (pos != key_lo(best.key[4]));
    tieb = tie ? d2b : tieb;
    if (inb & fresh & (key < best.key[4])) {
        const uint32_t out5 = key_hi(best.key[4]);        // the point this insertion pushes out of the set ...
        top5k_insert(best, key);
        tieb = (out5 == key_hi(best.key[4])) ? out5 : tieb;   // ... may stand at exactly the new 5th distance
        bound = fminf(__uint_as_float(key_hi(best.key[4])), gatef);
    }
}

// The exact walk that settles a tie across the 5th/6th boundary in the fallback sweep (rare: two map points at
// bit-identical distance): the lane's 3x3x3 cells, keys (d2, ORIGINAL index), positions carried along.  Result as
// (d2 | position) keys in the reference order.
__device__ __forceinline__ void exact_top5_of_cells(gptr<const v4f> map, gptr<const int32_t> cell_start, const GridDesc& g,
                                                    int cx, int cy, int cz, float sx, float sy, float sz, float gatef, Top5k& out)
{
    uint64_t ek[5];
    int ep[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { ek[k] = kKeyInf; ep[k] = 0x7fffffff; }
    float bound = gatef;
#pragma unroll 1
    for (int k = 0; k < 9; k++) {
        const int yy = cy + run_dy(k), zz = cz + run_dz(k);
        if (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) continue;
        const int rb = (zz * g.ny + yy) * g.nx;
        const int js = cell_start[rb + max(cx - 1, 0)], je = cell_start[rb + min(cx + 1, g.nx - 1) + 1];
#pragma unroll 1
        for (int j = js; j < je; j++) {
            float d2;
            const uint64_t key = make_key(map[j], sx, sy, sz, d2);
            if (d2 <= bound && key < ek[4]) {
                ek[4] = key; ep[4] = j;
#pragma unroll
                for (int q = 4; q > 0; --q) {
                    const bool c = ek[q] < ek[q - 1];
                    const uint64_t tk = ek[q - 1]; ek[q - 1] = c ? ek[q] : tk; ek[q] = c ? tk : ek[q];
                    swap_if(c, ep[q - 1], ep[q]);
                }
                bound = fminf(__uint_as_float(key_hi(ek[4])), gatef);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) out.key[k] = (ek[k] == kKeyInf) ? kKeyInf : (((uint64_t)key_hi(ek[k]) << 32) | (uint32_t)ep[k]);
}

// A lane's own search over its 3x3x3 cells, every lane of the wave for itself - for the lanes of a pass that are too far apart
// to share a tile (a sparse far-field ring segment: consecutive points metres apart): all of them walk at once, each its <= 9
// short runs, instead of being served one after the other.  Two steps, so that the wave can look at the candidate counts
// before it commits: walk_runs sizes the runs (all eighteen cell_start reads in one round trip; the runs go to the lane's
// column of `lruns`, [k][lane]) and returns the lane's candidate count; walk_top7 then measures them, four loads in flight,
// and keeps the seven nearest as (d2 | position) keys, ascending.
// `lim2`: rows / cells whose slab bound lies beyond it are skipped (every map point nearer than that is still met).
__device__ __forceinline__ int walk_runs(gptr<const int32_t> cell_start, const GridDesc& g, int cx, int cy, int cz, float lim2,
                                         float gx2m, float gx2p, float gy2m, float gy2p, float gz2m, float gz2p, int lane, int2* lruns)
{
    int P = 0;
#pragma unroll 3
    for (int k = 0; k < 9; k++) {
        const int dyc = run_dy(k), dzc = run_dz(k);
        const int yy = cy + dyc, zz = cz + dzc;
        const float lb = (dyc < 0 ? gy2m : (dyc > 0 ? gy2p : 0.0f)) + (dzc < 0 ? gz2m : (dzc > 0 ? gz2p : 0.0f));
        int js = 0, je = 0;
        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz && !(lb > lim2)) {
            const int xs = (cx > 0 && lb + gx2m <= lim2) ? cx - 1 : cx;
            const int xe = (cx < g.nx - 1 && lb + gx2p <= lim2) ? cx + 1 : cx;
            const int rb = (zz * g.ny + yy) * g.nx;
            js = cell_start[rb + xs]; je = cell_start[rb + xe + 1];
        }
        lruns[k * 64 + lane] = make_int2(js, je);
        P += je - js;
    }
    return P;
}

// `own`: the lane whose runs are walked (its column of lruns); `part` / `team`: this lane takes candidates part, part + team, ...
// of every run (a team of lanes shares one searching point: the parts' lists are merged by the caller).
__device__ __forceinline__ void walk_top7(gptr<const v4f> map, float sx, float sy, float sz, int own, int part, int team, bool on,
                                          const int2* lruns, uint64_t (&f)[7])
{
#pragma unroll
    for (int k = 0; k < 7; k++) f[k] = kKeyInf | 0xffffffffull;
#pragma unroll 1
    for (int k = 0; k < 9; k++) {
        const int2 r = lruns[k * 64 + own];
        const int end = on ? r.y : r.x;
#pragma unroll 1
        for (int j = r.x + part; j < end; j += 2 * team) {
            const bool two = j + team < end;
            v4f m0 = map[j], m1 = map[two ? j + team : j];
            m0.w = __int_as_float(j); m1.w = __int_as_float(j + team);
            float d0, d1;
            const uint64_t k0 = make_key(m0, sx, sy, sz, d0);
            const uint64_t k1 = two ? make_key(m1, sx, sy, sz, d1) : (kKeyInf | 0xffffffffull);
            if (k0 < f[6] || k1 < f[6]) {
                f[6] = (k0 < f[6]) ? k0 : f[6];
#pragma unroll
                for (int q = 6; q > 0; --q) cas_u64(f[q - 1], f[q]);
                f[6] = (k1 < f[6]) ? k1 : f[6];
#pragma unroll
                for (int q = 6; q > 0; --q) cas_u64(f[q - 1], f[q]);
            }
        }
    }
}

// per-wave diagnostics of the hook variant
struct WaveProf {
    int mode = 0;              // search path taken by the last entry that searched: 1 tile, 2 served, 3 tile overflow then served
    int n_a = 0, n_b = 0, n_c = 0;   // lanes settled by certificate / by re-measuring / by searching (summed over the wave's entries)
    int rows = 0, pts = 0, raw = 0, why = 0;
    int n_fb = 0, reach_mm = 0, kq = 0, cmax = 0;   // tile path: lanes that fell back to the exact sweep, staged reach, lanes per point, longest list
    unsigned long long ts[4] = { 0, 0, 0, 0 };   // stamps of the last entry that searched: search begins, tile staged, tile done, search done
};

// ------------------------------------------------------------------------------------------
// pass 1: bring tuple / plane / certificate of the points of one wave-table entry up to date
// ------------------------------------------------------------------------------------------
// What a lane without slack needs to re-measure its six front members, requested by k_register while the previous
// iteration is being closed (none of it depends on the pose).  `on`: this lane's copy is valid.
struct Fragile {
    bool on = false;
    v4i aux = { 0, 0, 0, 0 };
    int opos[5] = { 0, 0, 0, 0, 0 };
    v4f fr[6];                                            // the front: x, y, z, position in map_sorted (bits)
    int alt[5] = { 0, 0, 0, 0, 0 };                       // the tuple the lane had before its present one, and that tuple's plane
    v4f alt_plane = { NAN, 0.0f, 0.0f, 0.0f };            // (valid if state bit 4 is set)
};

// The seven smallest 32-bit keys seen, ascending.  a[k-1] <= a[k] always, so the new a[k] is the median of the old
// a[k-1], the old a[k] and the key: the key if it falls between them, a[k-1] if it pushes the list down, a[k] otherwise -
// one v_med3_u32 per element, from the top down (each uses the old value below it), and a minimum for a[0].
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }
__device__ __forceinline__ void near7_insert(uint32_t (&a)[7], uint32_t key)
{
#pragma unroll
    for (int k = 6; k > 0; --k) a[k] = umed3(a[k - 1], a[k], key);
    a[0] = min(a[0], key);
}
constexpr uint32_t kSlotBits = 9;                          // a tile slot in the low bits of a sweep key (kTilePts <= 512)
constexpr uint32_t kSlotMask = (1u << kSlotBits) - 1u;
constexpr uint32_t kNoKey = 0xffffffffu;
static_assert(kTilePts <= (1 << kSlotBits), "tile slots must fit the key's low bits");

// LEAN: the 128-register builds (two workgroups per CU: the scan slots of a batch, the 16-wave shape of large scans).  There a
// pass stores its lanes' fronts straight from the tile and the plane fit fetches the five points again from the map
// (L2-warm), instead of carrying 24 registers of coordinates through the passes that may follow.
// WALK: the lanes of a scattered sparse wave may walk their own cells (not in the scan slots of a batch, whose kernel is
// measurably slower for every line of search code it carries).
template <bool HOOK, bool LEAN, bool WALK>
__device__ __forceinline__ void associate_chunk(CtxP cp, const GridDesc& g, gptr<const v4f> map,
                                                gptr<const int32_t> cell_start, const float (&T)[12], float gatef, int ablate,
                                                int2 chunk, int lane, v4f* lpts, int2* lrows, float* lkeep,
                                                float px, float py, float pz, v4f cert, WaveProf& prof)
{
    const int nq = cp->n_q;
    const int i = chunk.x + lane;
    const bool valid = lane < chunk.y && i < nq;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    if (valid) {
        // pointAssociateToMap (:302-308), association order of the reference expression
        sx = ((T[0] * px + T[1] * py) + T[2]  * pz) + T[3];
        sy = ((T[4] * px + T[5] * py) + T[6]  * pz) + T[7];
        sz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11];
    }
    const bool fin = valid && (fabsf(sx) < 3.0e38f) && (fabsf(sy) < 3.0e38f) && (fabsf(sz) < 3.0e38f);

    // ---- tier A: has the point moved less than its slack since its tuple was established?
    const float ex = sx - cert.x, ey = sy - cert.y, ez = sz - cert.z;
    const float eps = sqrtf((ex * ex + ey * ey) + ez * ez) * 1.0001f + 1e-6f;
    const bool passA = fin && !(ablate & 1) && (eps < cert.w);          // slack 0 (no certificate) and NaN never pass
    const bool need = fin && !passA;
    if (HOOK) prof.n_a += __popcll(__ballot(passA));
    if (!HOOK && !__ballot(need)) return;                                // the whole entry is certified

    const auto auxp = G((v4i*)cp->aux);
    const auto nposp = G(cp->npos);
    const auto frontp = G((v4f*)cp->front);
    const auto certp = G((v4f*)cp->cert);
    const auto planep = G((v4f*)cp->plane_cache);
    const float gate_r = cp->gate_r;

    // ---- what the point remembers, all of it in one round trip: its tuple (for change detection), its front - the six
    // nearest map points it knew, with their coordinates - and the radius beyond which it knows nothing.  A point that has
    // never been settled (a new scan: certificate all zero) remembers nothing and asks for nothing.
    const bool fresh = cert.x == 0.0f && cert.y == 0.0f && cert.z == 0.0f && cert.w == 0.0f;
    const bool lookB = (HOOK ? fin : need) && !fresh;
    int ost = 0, nb_n = 0;
    float r_out = 0.0f;                                   // every map point other than the front members was at least this far from q_ref
    int opos[5] = { 0, 0, 0, 0, 0 };
    v4f fr[kNbr];
#pragma unroll
    for (int k = 0; k < kNbr; k++) fr[k] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
    if (lookB) {                                          // (meaningful only if the state word says the front is valid)
        const v4i a = auxp[i];
#pragma unroll
        for (int j = 0; j < 5; j++) opos[j] = nposp[(size_t)j * nq + i];
#pragma unroll
        for (int k = 0; k < kNbr; k++) fr[k] = frontp[(size_t)k * nq + i];
        ost = a.y; nb_n = min(max(a.z, 0), kNbr); r_out = fminf(__int_as_float(a.x), __int_as_float(a.w));
    }
    const bool had5 = lookB && (ost & 4) != 0;
    bool nbr_ok = lookB && (ost & 8) != 0;                // the stored front holds every map point within r_out of q_ref (possibly none)
    if (HOOK && valid && passA) {
        // certified lanes report the stored tuple in the STORED order with the distances measured now: a wrong
        // certificate shows up in the parity tests as a mis-ordered or wrong neighbour list
        const int o = G(cp->qperm)[i];
        const bool gatedA = (ost & 3) != 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const v4f m = map[had5 ? opos[j] : 0];
            float d2;
            make_key(m, sx, sy, sz, d2);
            if (cp->dbg_idx5) G(cp->dbg_idx5)[5 * (size_t)o + j] = gatedA ? __float_as_int(m.w) : -1;
            if (cp->dbg_d2) G(cp->dbg_d2)[5 * (size_t)o + j] = gatedA ? d2 : INFINITY;
        }
    }

    Top6k top;                                            // the outcome: the six nearest as (d2 | position), ascending
#pragma unroll
    for (int k = 0; k < 6; k++) top.key[k] = kKeyInf;
    float rn = 0.0f;                                      // every map point other than the six is at least this far away
    bool settled = false;                                 // the lane has its answer for this pose
    v4f nb[6];                                            // coordinates of the six (search: from the tile; w is not the original index there)
#pragma unroll
    for (int k = 0; k < 6; k++) nb[k] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
    bool have_nb = false;
    bool front_done = false;                              // (LEAN) the pass has stored this lane's front already

    // ---- tier B: the front, measured at this pose (no memory access: its coordinates came with the state).  Everything
    // else is at least r_out - (distance moved since) away: if that is beyond the new 5th member the tuple is proved; if it
    // is beyond the gate and fewer than 5 members are inside the gate, "not gated" is proved.  A lane it does not settle
    // takes the 5th distance along as the bound of its search.
    float bound = gatef;                                  // upper bound of the 5th-neighbour distance (squared), capped at the gate
    bool has_prior = false;
    {
        const bool ev = need && nbr_ok && !(ablate & 2);
        if (__ballot(ev)) {
            // (measured straight into the outcome; a lane the front does not settle starts its search from an empty list again)
#pragma unroll
            for (int k = 0; k < kNbr; k++) {
                const bool on = ev && k < nb_n;
                float d2;
                const uint64_t key = make_key(fr[k], sx, sy, sz, d2);
                top.key[k] = on ? key : kKeyInf;
            }
            sort6k(top);
            const float r = r_out - eps;
            const float d2_5 = __uint_as_float(key_hi(top.key[4]));
            const bool have5 = key_hi(top.key[4]) < 0x7f800000u;
            const bool in_gate = have5 && ((double)d2_5 < cp->gate_sq);
            const bool ok = in_gate ? (sqrtf(d2_5) + kCertMargin < r) : (gate_r + kCertMargin < r);
            if (ev && ok) { rn = r; settled = true; }
            else {
                if (ev && have5 && !(ablate & 16)) { bound = fminf(d2_5, gatef); has_prior = true; }
#pragma unroll
                for (int q = 0; q < 6; q++) top.key[q] = kKeyInf;
            }
        }
        if (HOOK) prof.n_b += __popcll(__ballot(settled));
    }

    // ================================ tier C: search =================================
    const bool searching = need && !settled;
    const unsigned long long cmask = __ballot(searching);
    if (cmask) {
        if (HOOK) { prof.n_c += __popcll(cmask); prof.ts[0] = wall_clock64(); }
        if (searching) { nb_n = 0; nbr_ok = false; }
        has_prior = has_prior && searching;
        const int cx = cell_coord(sx, g.ox, g.inv_e, g.nx);
        const int cy = cell_coord(sy, g.oy, g.inv_e, g.ny);
        const int cz = cell_coord(sz, g.oz, g.inv_e, g.nz);
        // squared slab distances of this query to the faces of its own cell: lower bounds of the
        // distance to anything in the neighbouring row / cell on that side (see kSlabMargin)
        // (measured from the grid origin, like the binning itself: the map lives in the odometry frame, and at 20 km from its
        // origin an absolute face coordinate would be rounded to a millimetre; what is left is the rounding of the binning,
        // which grows with the extent of the local map - see kSlabMargin)
        const float E = g.e;
        const float ux = sx - g.ox, uy = sy - g.oy, uz = sz - g.oz;
        const float slab = kSlabMargin + kSlabRound * (fabsf(ux) + fabsf(uy) + fabsf(uz));
        const float xlo = (float)cx * E, ylo = (float)cy * E, zlo = (float)cz * E;
        const float gxm = fmaxf(ux - xlo - slab, 0.0f), gxp = fmaxf(xlo + E - ux - slab, 0.0f);
        const float gym = fmaxf(uy - ylo - slab, 0.0f), gyp = fmaxf(ylo + E - uy - slab, 0.0f);
        const float gzm = fmaxf(uz - zlo - slab, 0.0f), gzp = fmaxf(zlo + E - uz - slab, 0.0f);
        const float gx2m = gxm * gxm * 0.9999f, gx2p = gxp * gxp * 0.9999f;
        const float gy2m = gym * gym * 0.9999f, gy2p = gyp * gyp * 0.9999f;
        const float gz2m = gzm * gzm * 0.9999f, gz2p = gzp * gzp * 0.9999f;
        // How far this lane looks: a reach beyond its bound, but not beyond what its 3x3x3 cells cover for certain
        // (one whole cell past the nearest face of its own cell).  Every map point nearer than `be` is met.
        const float cover = (E - slab) + fminf(fminf(fminf(gxm, gxp), fminf(gym, gyp)), fminf(gzm, gzp));
        const float sb = sqrtf(bound);
        const bool tight = has_prior && sb < kTightBound;
        const float be = searching ? fminf(sb + (tight ? kNbrReach : kNbrReachCold), cover) : 0.0f;
        const float be2 = fmaxf(be * be, bound);          // rows / cells are selected with this; never tighter than the bound itself
        const float rim2 = be * be * 0.99999f;            // a tile point nearer than this is one of EVERY map point nearer than this

        // ---- tile passes.  A pass takes the first `group` searching lanes that are still to do: the box rows those lanes need
        // are sized and streamed through a filter into the wave's LDS tile, and the lanes are settled from the tile.  A box far
        // larger than the lanes' own neighbourhoods (scattered points) is not worth staging: those lanes are served one by one
        // below.  When the tile of a pass would overflow (a dense part of the map under a wide wave), the pass is repeated with
        // half the lanes - consecutive lanes are neighbours in space, so the box and with it the tile shrink.
        unsigned long long todo = cmask, pend = 0ull, walk = 0ull;   // lanes to do in tile passes / to be served one by one / walking their own cells
        int group = 64;
        while (todo) {
        if (HOOK) prof.cmax++;                            // (diagnostics: pass attempts)
        const bool act = ((todo >> lane) & 1ull) != 0ull && __popcll(todo & ((1ull << lane) - 1ull)) < group;
        const unsigned long long amask = __ballot(act);
        const int nA = __popcll(amask);
        // bounding box of the pass's lanes
        const float mnx = wave_min_f32(act ? sx : INFINITY), mxx = wave_max_f32(act ? sx : -INFINITY);
        const float mny = wave_min_f32(act ? sy : INFINITY), mxy = wave_max_f32(act ? sy : -INFINITY);
        const float mnz = wave_min_f32(act ? sz : INFINITY), mxz = wave_max_f32(act ? sz : -INFINITY);
        // cell_coord is monotone: the box of the lanes' cells is the cells of the box corners
        const int bx0 = max(cell_coord(mnx, g.ox, g.inv_e, g.nx) - 1, 0), bx1 = min(cell_coord(mxx, g.ox, g.inv_e, g.nx) + 1, g.nx - 1);
        const int by0 = max(cell_coord(mny, g.oy, g.inv_e, g.ny) - 1, 0), by1 = min(cell_coord(mxy, g.oy, g.inv_e, g.ny) + 1, g.ny - 1);
        const int bz0 = max(cell_coord(mnz, g.oz, g.inv_e, g.nz) - 1, 0), bz1 = min(cell_coord(mxz, g.oz, g.inv_e, g.nz) + 1, g.nz - 1);
        const int nyb = by1 - by0 + 1, nzb = bz1 - bz0 + 1;
        const int R = nyb * nzb;                          // rows in the box
        if (HOOK) { prof.rows = R; prof.why = 0; }
        // (the filter box is compared in absolute coordinates: grown by their rounding as well)
        const float slabw = wave_max_f32(act ? slab + kAbsRound * (fabsf(sx) + fabsf(sy) + fabsf(sz)) : 0.0f);
        const int tile_cap = kTilePts - kTilePad;
        const float be2p = be2, rim2p = rim2;             // this pass's reach (squared) and rim: the lanes' own
        const float rr = sqrtf(wave_max_f32(act ? be2p : 0.0f)) * 1.000001f + slabw;
        const float fx0 = mnx - rr, fx1 = mxx + rr, fy0 = mny - rr, fy1 = mxy + rr, fz0 = mnz - rr, fz1 = mxz + rr;
        const bool few = nA <= cp->tune[3] && !(ablate & 128);      // a handful of lanes may be served one by one instead (experiments: S2M_TUNE)
        bool tile = !(ablate & 64) && !few && R <= kRowMax && (bx1 - bx0 + 1) * R <= 20 * max(nA, 8);
        const bool scattered = !tile && !few && !(ablate & 64);
        if (HOOK && !tile) prof.why = 1;
        int nt = 0;                                       // tile fill, wave-uniform
        // ---- the sweep of the tile as it stands: every lane keeps the seven nearest tile points by 32-bit keys - the bits of the
        // squared distance with the low kSlotBits replaced by the tile slot.  Such a key orders two points correctly unless
        // their distances agree to 2^-14 (18 um at 0.6 m); its distance part, the low bits cleared, is a LOWER bound of the
        // point's true squared distance.  Eight subtractions / multiplications / additions, one v_and_or and seven v_med3 /
        // v_min per lane and tile point, whatever the lane finds.
        // Idle lanes join in (`share`): with nA <= 32 searching lanes, kq = 2, 4 or 8 lanes share one searching point - the wave is
        // cut into kq parts of nslot = 64 / kq lanes, lane l works for slot l % nslot (the searching lane of that rank) and takes
        // every kq-th tile point; the parts' lists are merged by a butterfly, and the searching lane of rank r picks its result
        // up from lane r.  (The short waves of split chunks - 8, 16 or 32 points in a dense or a wide box - are the slowest
        // waves of a first launch.)
        auto sweep = [&](int ntile, bool share, uint32_t (&a)[7]) {
            wave_lds_sync();
            const int kq = (!share || nA > 32 || ntile < kShareMin) ? 1 : ((nA > 16) ? 2 : ((nA > 8) ? 4 : 8));
            const int nslot = 64 / kq, part = lane / nslot;
            int col = lane;                                   // the lane whose list is mine: my rank among the searching lanes
            float qx_ = sx, qy_ = sy, qz_ = sz;
            if (kq > 1) {
                int* lown = reinterpret_cast<int*>(lrows);    // (the row table is not needed any more when `share` is set)
                const int rank = __popcll(amask & ((1ull << lane) - 1ull));
                if (act) lown[rank] = lane;
                wave_lds_sync();
                const int s_ = lane & (nslot - 1);
                const int own = (s_ < nA) ? lown[s_] : lane;  // the searching lane this lane works for
                col = act ? rank : lane;
                qx_ = __shfl(sx, own, 64); qy_ = __shfl(sy, own, 64); qz_ = __shfl(sz, own, 64);
            }
            // behind the last tile point: entries far from everything, so that a step of four needs no bounds test
            if (lane < kTilePad) lpts[ntile + lane] = v4f{ 1.0e18f, 1.0e18f, 1.0e18f, 0.0f };
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 7; k++) a[k] = kNoKey;
            {
                // four tile points per step; the next step's reads are in flight behind this step's arithmetic
                const v4f* lp = lpts + part;
                int jv = part;
                v4f m0 = lp[0], m1 = lp[kq], m2 = lp[2 * kq], m3 = lp[3 * kq];
                for (int base = 0; base < ntile; base += 4 * kq) {
                    const v4f c0 = m0, c1 = m1, c2 = m2, c3 = m3;
                    lp += 4 * kq;
                    m0 = lp[0]; m1 = lp[kq]; m2 = lp[2 * kq]; m3 = lp[3 * kq];       // (at most 4 kq entries past the tile: the pad)
                    float d0, d1, d2_, d3;
                    make_key(c0, qx_, qy_, qz_, d0); make_key(c1, qx_, qy_, qz_, d1);
                    make_key(c2, qx_, qy_, qz_, d2_); make_key(c3, qx_, qy_, qz_, d3);
                    near7_insert(a, (__float_as_uint(d0) & ~kSlotMask) | (uint32_t)jv);
                    near7_insert(a, (__float_as_uint(d1) & ~kSlotMask) | (uint32_t)(jv + kq));
                    near7_insert(a, (__float_as_uint(d2_) & ~kSlotMask) | (uint32_t)(jv + 2 * kq));
                    near7_insert(a, (__float_as_uint(d3) & ~kSlotMask) | (uint32_t)(jv + 3 * kq));
                    jv += 4 * kq;
                }
            }
            for (int m = nslot; m < 64; m <<= 1) {            // merge the parts' lists
                uint32_t o[7];
#pragma unroll
                for (int k = 0; k < 7; k++) o[k] = (uint32_t)__shfl_xor((int)a[k], m, 64);
#pragma unroll
                for (int k = 0; k < 7; k++) near7_insert(a, o[k]);
            }
            if (kq > 1) {
#pragma unroll
                for (int k = 0; k < 7; k++) a[k] = (uint32_t)__shfl((int)a[k], col, 64);
            }
        };
        // ---- the six nearest by key, measured exactly and put in exact order (keys: distance bits | tile slot); beyond the
        // reach nothing is known, so a point farther than that is no member.  Every tile point other than the six has a key
        // >= a[6]: its true squared distance is at least a[6]'s distance part (returned).
        auto exact_six = [&](const uint32_t (&a)[7], Top6k& t) -> float {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const bool have = act && a[k] != kNoKey;
                const int jk = have ? (int)(a[k] & kSlotMask) : 0;
                float d2;
                make_key(lpts[jk], sx, sy, sz, d2);
                t.key[k] = (have && d2 <= rim2p) ? (((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)jk) : kKeyInf;
            }
            sort6k(t);
            return (a[6] != kNoKey) ? __uint_as_float(a[6] & ~kSlotMask) : INFINITY;
        };
        // A box that holds more points than the tile: the tile is swept whenever it is full, its six go into the lane's running
        // list (exact keys: distance | map position) and staging goes on into an empty tile.  `rest2` bounds everything seen
        // that is not in the running list: what a sweep left outside its six, and what the list pushed out.
        Top6k run;
#pragma unroll
        for (int k = 0; k < 6; k++) run.key[k] = kKeyInf;
        float rest2 = INFINITY;
        int nchunks = 0;
        auto flush_chunk = [&](int ntile, bool share) {
            uint32_t a[7];
            sweep(ntile, share, a);
            Top6k t;
            rest2 = fminf(rest2, exact_six(a, t));
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const bool on = key_hi(t.key[k]) < 0x7f800000u;
                const v4f m = lpts[on ? (int)key_lo(t.key[k]) : 0];
                const uint64_t out = top6k_insert(run, on ? (((uint64_t)key_hi(t.key[k]) << 32) | (uint32_t)__float_as_int(m.w)) : kKeyInf);
                rest2 = fminf(rest2, __uint_as_float(min(key_hi(out), 0x7f800000u)));
            }
            nchunks++;
            wave_lds_sync();                              // (staging goes on into the same tile)
        };
        for (int rg = 0; rg < R && tile; rg += 64) {
            // each lane sets the bits of the (<= 9) box rows it still needs; one OR-reduce
            unsigned long long want = 0ull;
            if (act) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int dyc = run_dy(k), dzc = run_dz(k);
                    const int yy = cy + dyc, zz = cz + dzc;
                    const float lb = (dyc < 0 ? gy2m : (dyc > 0 ? gy2p : 0.0f)) + (dzc < 0 ? gz2m : (dzc > 0 ? gz2p : 0.0f));
                    const int r = (zz - bz0) * nyb + (yy - by0) - rg;
                    if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz && !(lb > be2p) && r >= 0 && r < 64)
                        want |= 1ull << r;
                }
            }
            const uint32_t nlo = wave_or_u32((uint32_t)want), nhi = wave_or_u32((uint32_t)(want >> 32));
            const bool mine = (((lane < 32) ? (nlo >> lane) : (nhi >> (lane - 32))) & 1u) != 0u;
            int gs = 0, len = 0;                          // lane r: row rg + r of the box
            if (mine) {                                   // whole rows: a compact wave's box is narrow in x
                const int r = rg + lane;
                const int zq = r / nyb;
                const int gcell = ((bz0 + zq) * g.ny + by0 + (r - zq * nyb)) * g.nx;
                gs = cell_start[gcell + bx0];
                len = cell_start[gcell + bx1 + 1] - gs;
            }
            const int ptot = __builtin_amdgcn_readlane(wave_incl_scan_i32(len), 63);
            if (HOOK) prof.raw += ptot;
            // ---- stage through the filter: 16 lanes per row, 16 rows in flight per pass (four loads per lane); a point
            // enters the tile only if it lies inside the lanes' point box grown by the largest radius.
            // Rows nobody marked, and marked rows that are empty, are squeezed out first (through LDS): every
            // batch of rows costs a dependent round trip to the map whatever it holds.
            const int sub = lane >> 4, l16 = lane & 15;
            const unsigned long long nzrows = __ballot(len > 0);
            const int nr = __popcll(nzrows);
            if (len > 0) lrows[__popcll(nzrows & ((1ull << lane) - 1ull))] = make_int2(gs, len);
            wave_lds_sync();
            for (int cb = 0; cb < nr && tile; cb += 16) {
                int gsr[4], nrw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int c = cb + 4 * u + sub;
                    gsr[u] = 0; nrw[u] = 0;
                    if (c < nr) { const int2 rw = lrows[c]; gsr[u] = rw.x; nrw[u] = rw.y; }
                }
                const int npass = wave_max_i32(max(max(nrw[0], nrw[1]), max(nrw[2], nrw[3])));
                for (int k0 = 0; k0 < npass && tile; k0 += 16) {
                    const int k = k0 + l16;
                    v4f pv[4];
                    bool hv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        hv[u] = k < nrw[u];
                        pv[u] = v4f{ 0, 0, 0, 0 };
                        if (hv[u]) pv[u] = map[gsr[u] + k];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        pv[u].w = __int_as_float(gsr[u] + k);     // the tile keeps the map POSITION as the key's low word
                        const bool in = hv[u] && pv[u].x >= fx0 && pv[u].x <= fx1 && pv[u].y >= fy0 && pv[u].y <= fy1 && pv[u].z >= fz0 && pv[u].z <= fz1;
                        const unsigned long long mk = __ballot(in);
                        const int cn = __popcll(mk);
                        if (nt + cn > tile_cap) {
                            // the tile is full: sweep what it holds and go on into an empty one - or (128-register builds: no room for
                            // a running list) give the pass up and take fewer lanes
                            if (LEAN) { tile = false; if (HOOK) prof.why = 3; break; }
                            flush_chunk(nt, false); nt = 0;
                        }
                        if (in) lpts[nt + __popcll(mk & ((1ull << lane) - 1ull))] = pv[u];
                        nt += cn;
                    }
                }
            }
        }

        if (HOOK) prof.ts[1] = wall_clock64();
        if (!tile) {
            // too dense or too scattered for one tile: again with fewer lanes (half, or the next power of two below), or - a
            // handful of lanes - served one by one
            if (!few && nA > kServeLanes && !(ablate & 64)) {
                if (WALK && scattered && group == 64) {
                    // A box too wide for its lanes, at the first attempt.  Fewer lanes may share a pass (consecutive lanes are
                    // neighbours) - but the lanes of a sparse far-field ring segment are metres apart one by one, and end up served
                    // one after the other, 1.5 us each.  If every lane has few candidates in its own cells - fewer than serving
                    // them would cost - they walk them below instead, every lane for itself and all at once.
                    int2* lruns = reinterpret_cast<int2*>(lpts);
                    const int P = walk_runs(cell_start, g, cx, cy, cz, be2, gx2m, gx2p, gy2m, gy2p, gz2m, gz2p, lane, lruns);
                    wave_lds_sync();                      // (the tile area is the next pass's again)
                    if (wave_max_i32(act ? P : 0) < kWalkMax) { walk |= amask; todo &= ~amask; continue; }
                }
                group = (nA > 32) ? 32 : ((nA > 16) ? 16 : ((nA > 8) ? 8 : 4));
                continue;
            }
            pend |= amask; todo &= ~amask;
            continue;
        }
        todo &= ~amask;
        {
            if (HOOK) { prof.mode = 1; prof.pts += nt; }
            nt = __builtin_amdgcn_readfirstlane(nt);
            if (LEAN || __builtin_expect(nchunks == 0, 1)) {
                // the whole box in one tile (nearly every pass)
                uint32_t a[7];
                sweep(nt, true, a);
                Top6k t;
                const float lb7 = exact_six(a, t);
                // The answer stands if the exact 5th distance is below the bound of everything outside the six; otherwise (a
                // near-tie the keys cannot resolve: rare) the lane is served exactly below.
                const uint32_t h5 = key_hi(t.key[4]);
                const bool sure = !(h5 < 0x7f800000u) || (__uint_as_float(h5) < lb7);
                const unsigned long long unsure_m = __ballot(act && !sure);
                pend |= unsure_m;
                if (HOOK) prof.n_fb = __popcll(unsure_m);
                if (act && sure) {
                    int n6 = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const bool on = key_hi(t.key[k]) < 0x7f800000u;
                        const v4f m = lpts[on ? (int)key_lo(t.key[k]) : 0];
                        top.key[k] = on ? (((uint64_t)key_hi(t.key[k]) << 32) | (uint32_t)__float_as_int(m.w)) : kKeyInf;
                        n6 += on ? 1 : 0;
                        if (LEAN) { if (on) frontp[(size_t)k * nq + i] = m; }       // (a tile entry is a front entry: x, y, z, map position)
                        else nb[k] = on ? m : v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
                    }
                    nb_n = n6; nbr_ok = true; settled = true; have_nb = !LEAN; front_done = LEAN;
                    rn = sqrtf(fminf(lb7, rim2p)) * 0.999999f;
                }
            } else {
                // the last tile of several: the running list holds the answer if its 5th is nearer than everything left outside
                flush_chunk(nt, false);
                const uint32_t h5 = key_hi(run.key[4]);
                const bool sure = !(h5 < 0x7f800000u) || (__uint_as_float(h5) < rest2);
                const unsigned long long unsure_m = __ballot(act && !sure);
                pend |= unsure_m;
                if (HOOK) prof.n_fb = __popcll(unsure_m);
                if (act && sure) {
                    int n6 = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) { top.key[k] = run.key[k]; n6 += (key_hi(run.key[k]) < 0x7f800000u) ? 1 : 0; }
                    nb_n = n6; nbr_ok = true; settled = true; have_nb = false;       // (the coordinates come from the map again)
                    rn = sqrtf(fminf(rest2, rim2p)) * 0.999999f;
                }
            }
            wave_lds_sync();                              // (the next pass stages into the same tile)
        }
        }   // tile passes
        if (HOOK) prof.ts[2] = wall_clock64();
        if (pend) {
            if (HOOK) prof.mode = (prof.why >= 2) ? 3 : 2;
            uint64_t* lkeys = reinterpret_cast<uint64_t*>(lpts);
            constexpr int kServeCap = kTilePts * 2;           // u64 keys that fit the tile area
            while (pend) {
                const int L = (int)__builtin_ctzll(pend);
                pend &= pend - 1ull;
                const float qsx = lane_bcast(sx, L), qsy = lane_bcast(sy, L), qsz = lane_bcast(sz, L);
                const float qe2 = lane_bcast(be2, L);
                const int qcx = __builtin_amdgcn_readlane(cx, L), qcy = __builtin_amdgcn_readlane(cy, L), qcz = __builtin_amdgcn_readlane(cz, L);
                const float qy2m = lane_bcast(gy2m, L), qy2p = lane_bcast(gy2p, L), qz2m = lane_bcast(gz2m, L), qz2p = lane_bcast(gz2p, L);
                const float qx2m = lane_bcast(gx2m, L), qx2p = lane_bcast(gx2p, L);
                int rs = 0, rl = 0;
                if (lane < 9) {
                    const int dyc = run_dy(lane), dzc = run_dz(lane);
                    const int yy = qcy + dyc, zz = qcz + dzc;
                    const float lb = (dyc < 0 ? qy2m : (dyc > 0 ? qy2p : 0.0f)) + (dzc < 0 ? qz2m : (dzc > 0 ? qz2p : 0.0f));
                    if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz && !(lb > qe2)) {
                        const int xs = (qcx > 0 && lb + qx2m <= qe2) ? qcx - 1 : qcx;
                        const int xe = (qcx < g.nx - 1 && lb + qx2p <= qe2) ? qcx + 1 : qcx;
                        const int rb = (zz * g.ny + yy) * g.nx;
                        rs = cell_start[rb + xs];
                        rl = cell_start[rb + xe + 1] - rs;
                    }
                }
                const int incl = wave_incl_scan_i32(rl);
                const int P = __builtin_amdgcn_readlane(incl, 63);
                if (P > kServeCap) { walk |= 1ull << L; continue; }                   // more candidates than the LDS area holds
                int rst[9], rpe[9];
#pragma unroll
                for (int k = 0; k < 9; k++) { rst[k] = __builtin_amdgcn_readlane(rs, k); rpe[k] = __builtin_amdgcn_readlane(incl, k); }
                for (int base = 0; base < P; base += 64) {
                    const int t = base + lane;
                    int off = rst[0];
#pragma unroll
                    for (int k = 1; k < 9; k++) off = (t >= rpe[k - 1]) ? (rst[k] - rpe[k - 1]) : off;
                    if (t < P) {
                        v4f m = map[t + off];
                        m.w = __int_as_float(t + off);
                        float d2;
                        lkeys[t] = make_key(m, qsx, qsy, qsz, d2);
                    }
                }
                if (HOOK) prof.pts += P;
                wave_lds_sync();
                uint64_t last = 0ull, found[7];
#pragma unroll
                for (int r = 0; r < 7; r++) {
                    uint64_t mine = kKeyInf | 0xffffffffull;                 // above every real key
                    for (int t = lane; t < P; t += 64) {
                        const uint64_t k = lkeys[t];
                        const bool cnd = (r == 0 || k > last) && k < mine;
                        mine = cnd ? k : mine;
                    }
                    // 64-bit minimum over the wave: distance word first, position word among its holders
                    const uint32_t hmin = wave_min_u32(key_hi(mine));
                    const uint32_t lmin = wave_min_u32(key_hi(mine) == hmin ? key_lo(mine) : 0xffffffffu);
                    last = ((uint64_t)hmin << 32) | lmin;
                    found[r] = last;
                }
                if (lane == L) {
                    // members: the nearest six that lie inside what the walked cells cover for certain; the radius: the 7th, or that rim
                    int n6 = 0;
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const bool have = key_hi(found[r]) < 0x7f800000u && __uint_as_float(key_hi(found[r])) <= rim2;
                        top.key[r] = have ? found[r] : kKeyInf;
                        n6 += have ? 1 : 0;
                    }
                    const bool seven = n6 == 6 && key_hi(found[6]) < 0x7f800000u;
                    rn = sqrtf(fminf(seven ? __uint_as_float(key_hi(found[6])) : INFINITY, rim2)) * 0.999999f;
                    nb_n = n6; nbr_ok = true; settled = true; have_nb = false;
                }
                wave_lds_sync();                                  // the next lane's keys go to the same LDS area
            }
        }
        // ---- lanes that walk their own 3x3x3 cells, all at once: the lanes of passes too scattered to stage, and lanes with more
        // candidates than the served path holds
        const bool walker = searching && ((walk >> lane) & 1ull) != 0ull;     // (batch-slot builds: only lanes the served path could not hold)
        if (walk) {
            if (HOOK) prof.mode = 3;
            int2* lruns = reinterpret_cast<int2*>(lpts);
            (void)walk_runs(cell_start, g, cx, cy, cz, be2, gx2m, gx2p, gy2m, gy2p, gz2m, gz2p, lane, lruns);
            // Idle lanes join in: with nW <= 32 walking lanes, teams of 2, 4 or 8 lanes share one walking point - lane l works for
            // the walking lane of rank l % nslot and takes every team-th candidate of its runs; the parts' lists are merged by a
            // butterfly, and the walking lane of rank r picks its result up from lane r.  (The waves that walk are the short ones
            // of split chunks in the sparse far field: 8 or 16 points, the rest of the wave idle - and a walk is latency-bound.)
            const int nW = __popcll(walk);
            const int team = (nW > 32) ? 1 : ((nW > 16) ? 2 : ((nW > 8) ? 4 : 8));
            const int nslot = 64 / team, part = lane / nslot, s_ = lane & (nslot - 1);
            int* lown = reinterpret_cast<int*>(lrows);
            const int rank = __popcll(walk & ((1ull << lane) - 1ull));
            if (walker) lown[rank] = lane;
            wave_lds_sync();
            const bool on = s_ < nW;
            const int own = on ? lown[s_] : lane;
            const int col = walker ? rank : lane;
            const float qx_ = __shfl(sx, own, 64), qy_ = __shfl(sy, own, 64), qz_ = __shfl(sz, own, 64);
            uint64_t found[7];
            walk_top7(map, qx_, qy_, qz_, own, part, team, on, lruns, found);
            for (int m = nslot; m < 64; m <<= 1) {            // merge the parts' lists
                uint64_t o[7];
#pragma unroll
                for (int k = 0; k < 7; k++)
                    o[k] = ((uint64_t)(uint32_t)__shfl_xor((int)key_hi(found[k]), m, 64) << 32) | (uint32_t)__shfl_xor((int)key_lo(found[k]), m, 64);
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    if (o[k] < found[6]) {
                        found[6] = o[k];
#pragma unroll
                        for (int q = 6; q > 0; --q) cas_u64(found[q - 1], found[q]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 7; k++)
                found[k] = ((uint64_t)(uint32_t)__shfl((int)key_hi(found[k]), col, 64) << 32) | (uint32_t)__shfl((int)key_lo(found[k]), col, 64);
            if (walker) {
                int n6 = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const bool have = key_hi(found[r]) < 0x7f800000u && __uint_as_float(key_hi(found[r])) <= rim2;
                    top.key[r] = have ? found[r] : kKeyInf;
                    n6 += have ? 1 : 0;
                }
                const bool seven = n6 == 6 && key_hi(found[6]) < 0x7f800000u;
                rn = sqrtf(fminf(seven ? __uint_as_float(key_hi(found[6])) : INFINITY, rim2)) * 0.999999f;
                nb_n = n6; nbr_ok = true; settled = true; have_nb = false;
            }
            wave_lds_sync();
        }
        if (HOOK) prof.ts[3] = wall_clock64();
    }

    // ---- the outcome for every lane that re-measured or searched
    const bool upd = need && settled;
    uint64_t fk[6];
#pragma unroll
    for (int k = 0; k < 6; k++) fk[k] = top.key[k];
    bool tie56 = false;
    {   // The 5th and the 6th at bit-identical distance: more points may stand at that distance than the six kept, and the
        // reference order among them is by ORIGINAL index - the exact walk over the lane's cells decides (rare).
        const bool btie = upd && key_hi(fk[4]) < 0x7f800000u && key_hi(fk[4]) == key_hi(fk[5]);
        if (__ballot(btie)) {
            if (btie) {
                Top5k ex;
                exact_top5_of_cells(map, cell_start, g, cell_coord(sx, g.ox, g.inv_e, g.nx), cell_coord(sy, g.oy, g.inv_e, g.ny),
                                    cell_coord(sz, g.oz, g.inv_e, g.nz), sx, sy, sz, gatef, ex);
                if (ex.key[4] != kKeyInf) {
#pragma unroll
                    for (int k = 0; k < 5; k++) fk[k] = ex.key[k];
                }
                fk[5] = kKeyInf;                           // (whoever is sixth stands at the 5th distance: no slack, see below)
                have_nb = false;                           // (the coordinates in hand belong to the six the sweep found)
                if (searching) nbr_ok = false;             // no front is kept for such a point: it searches again next launch
            }
        }
        tie56 = btie;
    }
    const bool complete = upd && key_hi(fk[4]) < 0x7f800000u;
    bool changed = !had5;
#pragma unroll
    for (int k = 0; k < 5; k++) changed = changed || ((int)key_lo(fk[k]) != opos[k]);
    bool tie_adj = false;                                 // equal distances among the six: the reference order is (d2, ORIGINAL index)
#pragma unroll
    for (int k = 0; k < 5; k++) tie_adj = tie_adj || (key_hi(fk[k]) == key_hi(fk[k + 1]) && key_hi(fk[k]) < 0x7f800000u);
    tie_adj = tie_adj && upd;
    const bool store_front = upd && searching && nbr_ok && !front_done;  // a search's six become the point's front
    // Coordinates + original indices of the six nearest from the map (L2-warm) - only where they are needed and not in hand:
    // the plane has to be fitted (new tuple, or a tuple that was not gated before) or the front stored by a lane that was not
    // settled from a tile, or equal distances have to be put in index order (the tile carries positions, not original
    // indices).  A lane that re-measured and found its tuple unchanged keeps its plane and skips the round trip.
    const bool want_nb = upd && (changed || store_front || (ablate & 32) != 0 ||
                                 (complete && ((double)__uint_as_float(key_hi(fk[4])) < cp->gate_sq) && (ost & 3) == 0));
    const bool fetch6 = upd && (HOOK || tie_adj || (want_nb && !have_nb));
    if (__ballot(fetch6)) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const v4f m = map[(fetch6 && key_hi(fk[k]) < 0x7f800000u) ? (int)key_lo(fk[k]) : 0];
            nb[k] = fetch6 ? m : nb[k];
        }
    }
    {
        if (__ballot(tie_adj)) {
            if (tie_adj) {
#pragma unroll
                for (int pass = 0; pass < 5; pass++)
#pragma unroll
                    for (int k = 0; k < 5 - pass; k++) {
                        const bool c = key_hi(fk[k]) == key_hi(fk[k + 1]) && key_hi(fk[k]) < 0x7f800000u &&
                                       __float_as_int(nb[k].w) > __float_as_int(nb[k + 1].w);
                        const uint64_t tk = fk[k]; fk[k] = c ? fk[k + 1] : tk; fk[k + 1] = c ? tk : fk[k + 1];
                        const v4f tn = nb[k]; nb[k] = c ? nb[k + 1] : tn; nb[k + 1] = c ? tn : nb[k + 1];
                    }
                changed = !had5;
#pragma unroll
                for (int k = 0; k < 5; k++) changed = changed || ((int)key_lo(fk[k]) != opos[k]);
            }
        }
    }

    if (upd) {
        const float d2_5 = __uint_as_float(key_hi(fk[4]));
        const bool gated = complete && ((double)d2_5 < cp->gate_sq);                  // :1097
        v4f old_plane = { NAN, 0.0f, 0.0f, 0.0f };
        // (only a lane that re-measured: a lane that had to search moved far, and its old tuple is not coming back)
        const bool leaves = complete && changed && had5 && (ost & 3) != 0 && !HOOK && !searching;
        if (leaves) old_plane = planep[i];                                            // the plane of the tuple the lane leaves
        // The LS plane and its inlier test depend only on the ordered neighbour tuple, not on the pose: a point
        // that kept its tuple keeps its plane (and its verdict) bit for bit.
        int pst = (!changed && !(ablate & 32)) ? (ost & 3) : 0;
        if (gated && pst == 0) {
            float qr[5][3];
#pragma unroll
            for (int j = 0; j < 5; j++) { qr[j][0] = nb[j].x; qr[j][1] = nb[j].y; qr[j][2] = nb[j].z; }   // :1099-1101
            float X[3];
            plane_fit_5x3(qr, X);                                                    // :1104
            float pa = X[0], pb = X[1], pc = X[2], pd = 1.0f;
            const float ps = sqrtf(pa * pa + pb * pb + pc * pc);                      // :1111
            pa /= ps; pb /= ps; pc /= ps; pd /= ps;
            bool planeValid = true;
#pragma unroll
            for (int j = 0; j < 5; j++) {                                            // :1115-1122
                const float r = pa * nb[j].x + pb * nb[j].y + pc * nb[j].z + pd;
                if ((double)fabsf(r) > cp->plane_tol) planeValid = false;
            }
            const v4f pl = { planeValid ? pa : NAN, pb, pc, pd };                     // pa = NaN: contributes nothing
            planep[i] = pl;
            if (lkeep) { lkeep[3 * 64 + lane] = pl.x; lkeep[4 * 64 + lane] = pl.y; lkeep[5 * 64 + lane] = pl.z; lkeep[6 * 64 + lane] = pl.w; }
            pst = planeValid ? 1 : 2;
        } else if (!gated) {
            const v4f pl = { NAN, 0.0f, 0.0f, 0.0f };
            planep[i] = pl;
            if (lkeep) { lkeep[3 * 64 + lane] = pl.x; lkeep[4 * 64 + lane] = pl.y; lkeep[5 * 64 + lane] = pl.z; lkeep[6 * 64 + lane] = pl.w; }
            pst = 0;
        }
        // the certificate for the launches to come: d1..d5, then the nearest of anything else - the 6th of the six or
        // whatever lies beyond them
        float slack = 0.0f;
        const float d6 = (key_hi(fk[5]) < 0x7f800000u) ? fminf(sqrtf(__uint_as_float(key_hi(fk[5]))), rn) : rn;
        if (gated) {
            float dk[5];
#pragma unroll
            for (int k = 0; k < 5; k++) dk[k] = sqrtf(__uint_as_float(key_hi(fk[k])));
            float gap = d6 - dk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) gap = fminf(gap, dk[k + 1] - dk[k]);
            slack = fminf(0.5f * gap - kCertMargin, gate_r - dk[4] - 2.0f * kCertMargin);
        } else {
            // not gated: stays so while the 5th nearest map point (a member beyond the gate, or something outside the
            // six) cannot come inside the gate
            const float f5 = complete ? fminf(sqrtf(d2_5), rn) : rn;
            slack = f5 - gate_r - 2.0f * kCertMargin;
        }
        slack = (slack > 0.0f && !tie56) ? slack : 0.0f;                              // also NaN -> 0
        // (a certificate is never all zero - that is the mark of a point that has never been settled)
        const v4f cnew = { sx, sy, sz, (slack == 0.0f && sx == 0.0f && sy == 0.0f && sz == 0.0f) ? -1.0f : slack };
        certp[i] = cnew;
        // The tuple a lane leaves is kept with its plane (plane_alt / npos_alt, state bit 4): two nearly equidistant neighbours swap
        // places back and forth with the micro-steps of the converged loop, and the certify kernels exchange the two planes
        // instead of sending the lane here again.  (A kept pair stays valid whatever happens to the lane: a plane depends on
        // nothing but its tuple.)
        const bool keep_old = leaves;
        if (keep_old) {
            G((v4f*)cp->plane_alt)[i] = old_plane;
#pragma unroll
            for (int k = 0; k < 5; k++) G(cp->npos_alt)[(size_t)k * nq + i] = opos[k];
        }
        if (store_front) {
#pragma unroll
            for (int k = 0; k < kNbr; k++) {
                const v4f f = { nb[k].x, nb[k].y, nb[k].z, __int_as_float((int)key_lo(fk[k])) };
                if (k < nb_n) frontp[(size_t)k * nq + i] = f;
            }
        }
        const v4i anew = { __float_as_int(rn), pst | (complete ? 4 : 0) | (nbr_ok ? 8 : 0) | (keep_old ? 16 : (ost & 16)), nbr_ok ? nb_n : 0,
                           __float_as_int(rn) };
        auxp[i] = anew;
        if (complete && changed) {
#pragma unroll
            for (int k = 0; k < 5; k++) nposp[(size_t)k * nq + i] = (int)key_lo(fk[k]);
        }
        if (HOOK) {
            const int o = G(cp->qperm)[i];
            if (cp->dbg_idx5) {
#pragma unroll
                for (int j = 0; j < 5; j++) G(cp->dbg_idx5)[5 * (size_t)o + j] = gated ? __float_as_int(nb[j].w) : -1;
            }
            if (cp->dbg_d2) {
#pragma unroll
                for (int j = 0; j < 5; j++) G(cp->dbg_d2)[5 * (size_t)o + j] = __uint_as_float(key_hi(fk[j]));
            }
        }
    } else if (HOOK && valid && !fin) {
        const int o = G(cp->qperm)[i];
        if (cp->dbg_idx5) {
#pragma unroll
            for (int j = 0; j < 5; j++) G(cp->dbg_idx5)[5 * (size_t)o + j] = -1;
        }
        if (cp->dbg_d2) {
#pragma unroll
            for (int j = 0; j < 5; j++) G(cp->dbg_d2)[5 * (size_t)o + j] = INFINITY;
        }
    }
}

// residual, weight (:1125-1139) and Jacobian row (:1216-1234) of one point against its plane: keep = the point is a
// correspondence (laserCloudOriFlag); cf = coeffSel; row / rhs = its line of matA / matB
__device__ __forceinline__ bool linearise_row(CtxP cp, const float (&T)[12], const float (&sc6)[6],
                                              float px, float py, float pz, const v4f pl, float (&cf)[4], float (&row)[6], float& rhs)
{
    const float sx = ((T[0] * px + T[1] * py) + T[2]  * pz) + T[3];
    const float sy = ((T[4] * px + T[5] * py) + T[6]  * pz) + T[7];
    const float sz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11];
    const bool fin = (fabsf(sx) < 3.0e38f) && (fabsf(sy) < 3.0e38f) && (fabsf(sz) < 3.0e38f);
    bool keep = false;
    cf[0] = 0.0f; cf[1] = 0.0f; cf[2] = 0.0f; cf[3] = 0.0f;
    if (fin && pl.x == pl.x) {                                                        // gated, plane passed the inlier test
        const float pa = pl.x, pb = pl.y, pc = pl.z, pd = pl.w;
        const float pd2 = pa * sx + pb * sy + pc * sz + pd;                           // :1125
        const float rr = sqrtf(sqrtf(px * px + py * py + pz * pz));
        const float sw = (float)(1.0 - cp->weight_scale * (double)fabsf(pd2) / (double)rr);   // :1127
        if ((double)sw > cp->weight_min) {                                            // :1135
            cf[0] = sw * pa; cf[1] = sw * pb; cf[2] = sw * pc; cf[3] = sw * pd2;      // :1130-1133
            keep = true;
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) row[a] = 0.0f;
    rhs = 0.0f;
    if (keep) jacobian_row(sc6, px, py, pz, cf, row, rhs);
    return keep;
}

// product number k of a correspondence's line (21 upper-triangular JtJ, 6 Jtr, the count), k a compile-time constant
template <int K>
__device__ __forceinline__ double line_product(const float (&row)[6], float rhs, bool keep)
{
    if (K == kAcc - 1) return keep ? 1.0 : 0.0;
    if (K >= 21) return (double)row[K - 21 < 6 ? K - 21 : 0] * (double)rhs;
    // (a, b), a <= b, is number a*6 - a(a-1)/2 + (b-a)
    constexpr int a = K < 6 ? 0 : (K < 11 ? 1 : (K < 15 ? 2 : (K < 18 ? 3 : (K < 20 ? 4 : 5))));
    constexpr int base = a * 6 - (a * (a - 1)) / 2;
    constexpr int b = a + (K - base);
    return (double)row[a < 6 ? a : 0] * (double)row[(b >= 0 && b < 6) ? b : 0];
}

// ------------------------------------------------------------------------------------------
// pass 2: residual, weight, Jacobian row and the 28 products of one wave-table entry (:1125-1139, :1216-1239)
// ------------------------------------------------------------------------------------------
template <bool HOOK>
__device__ __forceinline__ void linearise_point(CtxP cp, const float (&T)[12], const float (&sc6)[6],
                                                int i, float px, float py, float pz, const v4f pl, double (&acc)[kAcc])
{
    float cf[4], row[6], rhs;
    const bool keep = linearise_row(cp, T, sc6, px, py, pz, pl, cf, row, rhs);
    if (keep) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) acc[k++] += (double)row[a] * (double)row[b];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += (double)row[a] * (double)rhs;
        acc[27] += 1.0;
    }
    if (HOOK) {
        const int o = G(cp->qperm)[i];
        if (cp->dbg_flag) G(cp->dbg_flag)[o] = keep ? 1 : 0;
        if (cp->dbg_coeff) {
#pragma unroll
            for (int j = 0; j < 4; j++) G(cp->dbg_coeff)[4 * (size_t)o + j] = cf[j];
        }
    }
}

template <bool HOOK>
__device__ __forceinline__ void linearise_chunk(CtxP cp, const float (&T)[12], const float (&sc6)[6],
                                                int2 chunk, int lane, double (&acc)[kAcc])
{
    const int i = chunk.x + lane;
    if (!(lane < chunk.y && i < cp->n_q)) return;
    const float px = G(cp->qx)[i], py = G(cp->qy)[i], pz = G(cp->qz)[i];              // pointOri (:1085)
    const v4f pl = G((const v4f*)cp->plane_cache)[i];
    linearise_point<HOOK>(cp, T, sc6, i, px, py, pz, pl, acc);
}

// ------------------------------------------------------------------------------------------
// The registration kernels.  One source, three roles (MODE):
//
//   kFused    R  one launch = the whole iteration: certificate test, association of the lanes that fail it (tiers B / C) in
//                line, residuals, partial row; with solve_prev its prologue closes the iteration before.  The kernel of a
//                single small scan, where nothing else wants the CU and a second kernel per iteration only adds a boundary.
//   kCertify  C  the same without tiers B / C: no tile, no candidate lists, a quarter of the registers.  A workgroup in
//                which some lane needs more than its certificate (and is not settled by the in-line re-measurement) writes
//                NO partial row: it puts its number on the launch's worklist and ...
//   kSearch   S  ... one workgroup of the search kernel per worklist entry does that workgroup's iteration again, association
//                included, with a register budget of its own, and writes the row.  With kAll set (launch 0 of a scan: nothing
//                is certified yet) every workgroup is "on the list" and no C launch precedes it.
//
// Which kernel writes a workgroup's row makes no difference to the row: the row of workgroup b is always
//   sum over its CNW waves w, in order, of ( lane tree ( sum over the wave's entries, in order, of the lane's 28 products ) ),
// computed by the same functions below (wave_reduce_acc, write_partial_row) - so a loop gives bitwise the same trace
// whether its lanes were certified, re-measured or searched, in one kernel or in two (tests/test_tiers_gpu.py).
//
// Several scans at once (s2m_optimize_batch): blockIdx.y selects the scan slot; the slots' loops advance in lockstep, one
// launch per iteration for all of them.
// ------------------------------------------------------------------------------------------
constexpr int kFused = 0, kCertify = 1, kSearch = 2;
constexpr int kCertifyWaves = 4;       // waves per SIMD the certify kernel is built for (8-wave workgroups) ...
constexpr int kCertifyWavesBig = 4;
constexpr int kCertifyLeanWaves = 8;   // the lean certify kernel: 64 registers
constexpr int kCertifyLeanEpw = 2;     // ... two entries per wave: four waves write a row of eight    // ... and in the 16-wave shape of large scans
constexpr int kFlagSolvePrev = 1;      // the prologue closes iteration launch-1 (fused loop)
constexpr int kFlagAll = 2;            // kSearch: every workgroup, no worklist (launch 0 of a scan, observation hooks)
constexpr int kFlagCloseAfter = 4;     // kSearch on a grid of ONE workgroup per slot: walk the whole worklist, then close this iteration (k_finalize's job) right here
// ---- wave reduction by recursive halving, 16 sums at a time: at mask m a lane keeps one half of its sums and hands the
// other half to lane^m, so 8+4+2+1 values cross instead of 4 x 16; after the four steps lane l holds, in a[0], sum number
// l>>2 over the 16 lanes that share its two low bits, and two full exchanges complete it.  Fixed order: bitwise
// reproducible.  The first two steps (12 of the 17 exchanges) are lane swaps in the vector unit, the rest go through LDS.  The 28 sums of the normal equations go through it in two batches of 14 (a batch is all a lean kernel
// needs to hold: 32 registers).
// one exchange of the halving at mask 32 / 16 without a shuffle: v_permlane32_swap trades the upper 32 lanes of its first
// operand for the lower 32 of its second (v_permlane16_swap: odd 16-lane rows for even ones), after which every lane holds
// the value it keeps in one register and the value it receives in the other - the same two numbers, the same one addition
template <int M>
__device__ __forceinline__ double halving_swap_add(double lo, double hi)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    const unsigned long long ul = (unsigned long long)__double_as_longlong(lo), uh = (unsigned long long)__double_as_longlong(hi);
    v2u_ w0, w1;
    if (M == 32) {
        w0 = __builtin_amdgcn_permlane32_swap((unsigned)ul, (unsigned)uh, false, false);
        w1 = __builtin_amdgcn_permlane32_swap((unsigned)(ul >> 32), (unsigned)(uh >> 32), false, false);
    } else {
        w0 = __builtin_amdgcn_permlane16_swap((unsigned)ul, (unsigned)uh, false, false);
        w1 = __builtin_amdgcn_permlane16_swap((unsigned)(ul >> 32), (unsigned)(uh >> 32), false, false);
    }
    const double x = __longlong_as_double((long long)(((unsigned long long)w1.x << 32) | w0.x));
    const double y = __longlong_as_double((long long)(((unsigned long long)w1.y << 32) | w0.y));
    return x + y;
}

__device__ __forceinline__ void wave_reduce16(double (&a)[16], int lane, double* red16 /* 16 slots of this wave's row */)
{
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = halving_swap_add<32>(a[j], a[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = halving_swap_add<16>(a[j], a[j + 4]);
#pragma unroll
    for (int h = 2, m = 8; h >= 1; h >>= 1, m >>= 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < h; j++) {
            double lo = a[j], hi = a[j + h];
            asm volatile("" : "+v"(lo), "+v"(hi));            // two values, not one dynamically indexed array element (a select chain per exchange)
            const double keepv = up ? hi : lo;
            const double sendv = up ? lo : hi;
            a[j] = keepv + __shfl_xor(sendv, m, 64);
        }
    }
    a[0] += __shfl_xor(a[0], 2, 64);
    a[0] += __shfl_xor(a[0], 1, 64);
    if ((lane & 3) == 0) red16[lane >> 2] = a[0];
}

constexpr int kAccHalf = kAcc / 2;     // sums per batch
__device__ __forceinline__ void wave_reduce_acc(const double (&acc)[kAcc], int lane, double* red_row /* [32] of this wave */)
{
#pragma unroll
    for (int half = 0; half < 2; half++) {
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = (k < kAccHalf) ? acc[half * kAccHalf + k] : 0.0;
        wave_reduce16(a, lane, red_row + 16 * half);
    }
}

// the sums of one correspondence line per lane (a wave with a single entry), without holding all 28 at once
template <int HALF>
__device__ __forceinline__ void wave_reduce_line(const float (&row)[6], float rhs, bool keep, int lane, double* red_row)
{
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = 0.0;
    // 0.0 + p, as the accumulators of the other kernels see it
    a[0] = 0.0 + line_product<HALF * kAccHalf + 0>(row, rhs, keep);   a[1] = 0.0 + line_product<HALF * kAccHalf + 1>(row, rhs, keep);
    a[2] = 0.0 + line_product<HALF * kAccHalf + 2>(row, rhs, keep);   a[3] = 0.0 + line_product<HALF * kAccHalf + 3>(row, rhs, keep);
    a[4] = 0.0 + line_product<HALF * kAccHalf + 4>(row, rhs, keep);   a[5] = 0.0 + line_product<HALF * kAccHalf + 5>(row, rhs, keep);
    a[6] = 0.0 + line_product<HALF * kAccHalf + 6>(row, rhs, keep);   a[7] = 0.0 + line_product<HALF * kAccHalf + 7>(row, rhs, keep);
    a[8] = 0.0 + line_product<HALF * kAccHalf + 8>(row, rhs, keep);   a[9] = 0.0 + line_product<HALF * kAccHalf + 9>(row, rhs, keep);
    a[10] = 0.0 + line_product<HALF * kAccHalf + 10>(row, rhs, keep); a[11] = 0.0 + line_product<HALF * kAccHalf + 11>(row, rhs, keep);
    a[12] = 0.0 + line_product<HALF * kAccHalf + 12>(row, rhs, keep); a[13] = 0.0 + line_product<HALF * kAccHalf + 13>(row, rhs, keep);
    wave_reduce16(a, lane, red_row + 16 * HALF);
}

// the workgroup's partial row: the sums of its CNW waves, in wave order (after a barrier)
template <int CNW>
__device__ __forceinline__ void write_partial_row(CtxP cp, int launch, int b, int tid, const double (*red)[32])
{
    if (tid < kAcc) {
        const auto partial_row = G(cp->partials) + ((size_t)(launch & 1) * (size_t)cp->nblocks + b) * kAcc;   // slot launch & 1
        const int at = tid < kAccHalf ? tid : 16 + (tid - kAccHalf);        // (the two batches of wave_reduce_acc)
        double s = red[0][at];
#pragma unroll
        for (int w = 1; w < CNW; w++) s += red[w][at];
        partial_row[tid] = s;
    }
}

// the certificate test of one lane at this pose (tier A)
__device__ __forceinline__ bool lane_needs(const float (&T)[12], float px, float py, float pz, const v4f cert, bool valid, int ablate,
                                           float& sx, float& sy, float& sz, float& eps)
{
    sx = ((T[0] * px + T[1] * py) + T[2]  * pz) + T[3];
    sy = ((T[4] * px + T[5] * py) + T[6]  * pz) + T[7];
    sz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11];
    const bool fin = valid && (fabsf(sx) < 3.0e38f) && (fabsf(sy) < 3.0e38f) && (fabsf(sz) < 3.0e38f);
    const float ex = sx - cert.x, ey = sy - cert.y, ez = sz - cert.z;
    eps = sqrtf((ex * ex + ey * ey) + ez * ez) * 1.0001f + 1e-6f;
    return fin && !(!(ablate & 1) && (eps < cert.w));
}

// ------------------------------------------------------------------------------------------
// k_certify_lean: the certify role for loops whose iterations are closed by k_finalize (the lockstep loop of a batch: its
// grid is several rounds of workgroups deep, and how many of them a CU holds at once is what bounds a steady launch).
// No close, no tile, one wave-table entry per wave, the sums formed 14 at a time: 64 registers, eight waves per SIMD.
// A lane that fails its certificate is re-measured in line against its six front members (as the fused kernel does for
// lanes without slack, but fetched on demand); a flip back to the previous order exchanges the kept planes.  Whatever that
// does not settle - and any partition with more than one entry per wave - sends the workgroup to the search kernel.
// ------------------------------------------------------------------------------------------
// (EPW entries per wave: a workgroup of NW / EPW waves writes the row of NW entries - every entry reduced on its own, the
// row summed in entry order, exactly as a workgroup of NW waves would - so that twice as many rows are in flight per CU.
// Only EPW = 1 is instantiated: EPW = 2 was measured slower - the rolled loop over the entries costs more instructions than
// the second round of workgroups costs time - and needed 24 bytes of scratch.)
template <int NW, int EPW, int MINW>
__global__ __launch_bounds__(NW / EPW * 64, MINW) void k_certify_lean(const SlotTable tbl, int launch)
{
    static_assert(NW % EPW == 0, "whole entries per wave");
    constexpr int NWL = NW / EPW;                              // waves of this workgroup
    const CtxP cp = ctx_const(tbl.ctx[blockIdx.y]);
    const auto st = G(tbl.st[blockIdx.y]);
    const int done = st->done, n_waves = st->n_waves;
    if (done) return;
    __shared__ double red[NW][32];
    __shared__ float park[EPW == 2 ? NWL : 1][7][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblocks = cp->nblocks;
    const int nb_act = min((n_waves + NW - 1) / NW, nblocks);
    const int b = (int)blockIdx.x;
    if (b >= nb_act) return;
    const int nq = cp->n_q;
    const int ablate = cp->ablate;
    const bool single = nb_act * NW >= n_waves;                // every wave of the partition has one entry at most
    // everything the entries need is requested together (the second entry's plane only once the first is under way: registers);
    // the entries are then taken one after the other by the same code (cur = the entry in hand)
    static_assert(EPW == 1 || EPW == 2, "one or two entries per wave");
    const auto tb = G((const int2*)cp->wave_table);
    int2 chunkA = make_int2(0, 0), chunkB = make_int2(0, 0);
    {
        const int eA = wave * nb_act + b, eB = (wave + NWL) * nb_act + b;
        if (eA < n_waves) { chunkA.x = tb[eA].x; chunkA.y = tb[eA].y; }
        if (EPW == 2 && eB < n_waves) { chunkB.x = tb[eB].x; chunkB.y = tb[eB].y; }
        chunkA.x = __builtin_amdgcn_readfirstlane(chunkA.x); chunkA.y = __builtin_amdgcn_readfirstlane(chunkA.y);
        chunkB.x = __builtin_amdgcn_readfirstlane(chunkB.x); chunkB.y = __builtin_amdgcn_readfirstlane(chunkB.y);
    }
    float px = 0.0f, py = 0.0f, pz = 0.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
    v4f cert = { 0, 0, 0, 0 }, certB = { 0, 0, 0, 0 }, plane0 = { NAN, 0, 0, 0 };
    if (lane < chunkA.y && chunkA.x + lane < nq) {
        const int i = chunkA.x + lane;
        px = G(cp->qx)[i]; py = G(cp->qy)[i]; pz = G(cp->qz)[i];                      // pointOri (:1085)
        cert = G((const v4f*)cp->cert)[i];
        plane0 = G((const v4f*)cp->plane_cache)[i];
    }
    if (EPW == 2 && lane < chunkB.y && chunkB.x + lane < nq) {
        const int i = chunkB.x + lane;
        bx = G(cp->qx)[i]; by = G(cp->qy)[i]; bz = G(cp->qz)[i];
        certB = G((const v4f*)cp->cert)[i];
    }
    // the transform of this launch's pose: left in the state block by k_finalize (or by the host for launch 0)
    float T[12], sc6[6];
    if (st->T_valid) {
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = st->T[k];
#pragma unroll
        for (int k = 0; k < 6; k++) sc6[k] = st->sc[k];
    } else {
        float pose[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pose[k] = st->pose2[launch & 1][k];
        build_transform(pose, lane, T, sc6);
    }
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(T[k])));
#pragma unroll
    for (int k = 0; k < 6; k++) sc6[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc6[k])));

    bool defer = !single;
    int2 chunk = chunkA;
#pragma unroll 1
    for (int q = 0; q < EPW && single; q++) {
        const int w = wave + q * NWL;                          // the wave of the row this entry belongs to
        const int e = w * nb_act + b;
        // (the products of the trig values in the Jacobian row are wave-uniform but live in vector registers - this part has no
        // scalar float arithmetic - and hoisted out of this loop they cost twenty registers: keep them where they are used)
#pragma unroll
        for (int k = 0; k < 6; k++) asm volatile("" : "+s"(sc6[k]));
        if (e >= n_waves) { if (lane < 32) red[w][lane] = 0.0; }
        else {
        const int i = chunk.x + lane;
        const bool valid0 = lane < chunk.y && i < nq;
        if (EPW == 2 && q == 1 && valid0) plane0 = G((const v4f*)cp->plane_cache)[i];
        float sx, sy, sz, eps;
        const bool need = lane_needs(T, px, py, pz, cert, valid0, ablate, sx, sy, sz, eps);
        bool bad = false;
        if (__builtin_expect(__ballot(need) != 0ull, 0)) {
            // in-line re-measurement of the six front members (see the fused kernel); nothing is written unless the order flipped
            // back to the tuple the lane had before.  (Rare: the second entry's point waits in LDS meanwhile, not in registers.)
            if (EPW == 2) {
                park[wave][0][lane] = bx; park[wave][1][lane] = by; park[wave][2][lane] = bz;
                park[wave][3][lane] = certB.x; park[wave][4][lane] = certB.y; park[wave][5][lane] = certB.z; park[wave][6][lane] = certB.w;
                asm volatile("" : "=v"(bx), "=v"(by), "=v"(bz));
            }
            bool okl = !need, flip = false;
            int nk[5] = { 0, 0, 0, 0, 0 };                    // the re-measured order (positions in map_sorted)
            v4i aux = { 0, 0, 0, 0 };
            if (need && !(ablate & 3)) {
                aux = G((const v4i*)cp->aux)[i];
                const int nfr = min(max(aux.z, 0), kNbr);
                okl = (aux.y & 12) == 12 && (aux.y & 3) != 0;      // tuple complete, neighbourhood valid, plane known
                Top6k t;
#pragma unroll
                for (int k = 0; k < 6; k++) t.key[k] = kKeyInf;
                {
                    v4f fr[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) fr[j] = G((const v4f*)cp->front)[(size_t)j * nq + i];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        float d2;
                        const uint64_t key = make_key(fr[k], sx, sy, sz, d2);            // L2_Simple order; low word: the member's map position
                        t.key[k] = (okl && k < nfr) ? key : kKeyInf;
                    }
                    sort6k(t);
                }
                const float r = __int_as_float(aux.w) - eps;                            // everything outside the six is at least this far away
                const float d2_5 = __uint_as_float(key_hi(t.key[4]));
                okl = okl && key_hi(t.key[4]) < 0x7f800000u && ((double)d2_5 < cp->gate_sq) && (sqrtf(d2_5) + kCertMargin < r);
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    okl = okl && key_hi(t.key[k]) != key_hi(t.key[k + 1]);
                    nk[k] = (int)key_lo(t.key[k]);
                }
                if (okl) {
#pragma unroll
                    for (int k = 0; k < 5; k++) flip = flip || nk[k] != G(cp->npos)[(size_t)k * nq + i];
                }
            }
            if (__ballot(flip)) {
                // the previous tuple's plane is kept: a flip back exchanges the two; any other order has to be fitted (search kernel)
                bool back = flip && (aux.y & 16) != 0;
                if (back) {
#pragma unroll
                    for (int k = 0; k < 5; k++) back = back && nk[k] == G(cp->npos_alt)[(size_t)k * nq + i];
                }
                if (back) {
                    const v4f pl_new = G((const v4f*)cp->plane_alt)[i];
                    G((v4f*)cp->plane_alt)[i] = plane0;                                 // what the lane had until now
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int o = G(cp->npos)[(size_t)k * nq + i];
                        G(cp->npos_alt)[(size_t)k * nq + i] = o;
                        G(cp->npos)[(size_t)k * nq + i] = nk[k];
                    }
                    G((v4f*)cp->plane_cache)[i] = pl_new;
                    plane0 = pl_new;
                    // the stored certificate spoke of the old order: none from now on
                    const v4f cnew = { cert.x, cert.y, cert.z, 0.0f };
                    G((v4f*)cp->cert)[i] = cnew;
                    const v4i anew = { aux.x, (aux.y & ~3) | ((pl_new.x == pl_new.x) ? 1 : 2) | 16, aux.z, aux.w };
                    G((v4i*)cp->aux)[i] = anew;
                }
                okl = okl && (!flip || back);
            }
            bad = __ballot(!okl) != 0ull;
            if (EPW == 2) {
                wave_lds_sync();
                bx = park[wave][0][lane]; by = park[wave][1][lane]; bz = park[wave][2][lane];
                certB.x = park[wave][3][lane]; certB.y = park[wave][4][lane]; certB.z = park[wave][5][lane]; certB.w = park[wave][6][lane];
            }
        }
        defer = defer || bad;
        if (!bad) {
            float cf[4], row[6], rhs;
            const v4f pl = valid0 ? plane0 : v4f{ NAN, 0, 0, 0 };
            const bool keep = linearise_row(cp, T, sc6, px, py, pz, pl, cf, row, rhs);   // (an idle lane: plane NaN, no line)
            wave_reduce_line<0>(row, rhs, keep, lane, red[w]);
            wave_reduce_line<1>(row, rhs, keep, lane, red[w]);
        }
        }
        // the second entry takes the first one's place
        chunk = chunkB; px = bx; py = by; pz = bz; cert = certB;
    }
    if (__syncthreads_or(defer ? 1 : 0)) {
        if (tid == 0) {
            const int slot = atomicAdd(cp->wl_count + (launch & 1), 1);
            G(cp->wl_items)[(size_t)(launch & 1) * (size_t)nblocks + slot] = b;
        }
        return;
    }
    write_partial_row<NW>(cp, launch, b, tid, red);
}

// NW waves per workgroup; MINW waves per SIMD the kernel is built for (the register budget: 512 / MINW per lane);
// CNW: waves per workgroup of the partition whose rows this kernel writes (kSearch: the C kernel's shape; otherwise NW)
template <bool HOOK, int NW, int MINW, int MODE, int CNW = NW>
__global__ __launch_bounds__(NW * 64, MINW) void k_register(const SlotTable tbl, int launch, int flags)
{
    static_assert(MODE == kSearch || CNW == NW, "only the search kernel works on another kernel's partition");
    static_assert(CNW % NW == 0, "a search workgroup takes the waves of a row in whole rounds");
    static_assert(!(HOOK && MODE == kCertify), "the observation hook needs the association in line");
    const CtxP cp = ctx_const(tbl.ctx[blockIdx.y]);
    // The loop state block never moves, so it comes with the kernel arguments: `done` and the wave count arrive
    // with the first round trip, in parallel with the DevCtx block, instead of behind a pointer chase.
    const auto st = G(tbl.st[blockIdx.y]);
    const int done = st->done, n_waves = st->n_waves;
    if (!HOOK && done) return;
    const bool solve_prev = MODE != kSearch && (flags & kFlagSolvePrev) != 0;
    unsigned long long tk_start = 0, clk1 = 0, clk2 = 0;
    unsigned long long lm_stamps[7] = { 0, 0, 0, 0, 0, 0, 0 };   // diagnostics of the fused LM close
    if (HOOK) tk_start = wall_clock64();

    constexpr bool kHasSearch = MODE != kCertify;
    constexpr int  kTileWaves = kHasSearch ? NW : 1;
    __shared__ v4f     s_pts[kTileWaves][kHasSearch ? kTilePts : 1];   // per wave: the tile, or (gather path) the lane's 9 (start, end) pairs
    static_assert(sizeof(v4f) * kTilePts >= sizeof(int32_t) * 18 * 64, "run table must fit the tile area");
    __shared__ double  red[CNW][32];
    __shared__ int2    s_rows[kTileWaves][kHasSearch ? 64 : 1];            // per wave: the non-empty box rows of the current row group
    constexpr bool kKeep = kHasSearch && NW == 8 && MINW == 2;             // (one 8-wave workgroup per CU: LDS to spare; two per CU, or the 16-wave shape, fill it with tiles)
    __shared__ float   s_keep[kKeep ? NW : 1][kKeep ? 7 : 1][kKeep ? 64 : 1]; // per wave: point + plane of the entry in hand, parked across the association
    __shared__ float   s_lm_out[8];                  // pose + loop-ended flag published by lm_close_iteration
    __shared__ LmShared s_lm;                        // (kCertify: the close has no tile area to borrow)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblocks = cp->nblocks;
    const int nb_act = min((n_waves + CNW - 1) / CNW, nblocks);            // workgroups the wave table needs
    // ---- which row of the partition this workgroup works on
    // (the search kernel over a worklist: workgroup j takes items j, j + gridDim.x, ... - late in a loop the list is short or
    // empty and a small grid costs less to launch than one workgroup per row)
    int b = (int)blockIdx.x;
    int wl_cnt = 0, wl_at = (int)blockIdx.x;
    const bool listed = MODE == kSearch && !(flags & kFlagAll);
    if (listed) {
        const auto wlc = G(cp->wl_count);
        wl_cnt = wlc[launch & 1];
        if (b == 0 && tid == 0) { wlc[(launch + 1) & 1] = 0; st->deferred_total += wl_cnt; }   // the next launch's list starts empty (nobody reads or appends to it now)
    } else if (MODE == kSearch) {
        if (b == 0 && tid == 0) G(cp->wl_count)[(launch + 1) & 1] = 0;
    }
    for (;;) {
    if (listed) {
        if (wl_at >= wl_cnt) {
            if (!(flags & kFlagCloseAfter)) return;
            break;                                                          // the list is done: close the iteration below
        }
        b = __builtin_amdgcn_readfirstlane(G(cp->wl_items)[(size_t)(launch & 1) * (size_t)nblocks + wl_at]);
    }
    if (b >= nb_act) return;                                                // the rest of the (fixed, graph-captured) grid idles
    // Wave w of workgroup b starts at entry w*nb_act + b of the wave table and advances by the number of waves in
    // the partition: neighbouring entries (similar cost: the sort runs from the dense near field to the sparse far
    // field) land on different CUs, which evens out both the work and the L2-miss queues.
    const int estride = nb_act * CNW;
    const auto tb = G((const int2*)cp->wave_table);
    const int nq = cp->n_q;
    const GridDesc g = grid_of(cp);
    const auto map = G((const v4f*)cp->map_sorted);
    const auto cell_start = G(cp->cell_start);
    const int ablate = cp->ablate;
    const float gatef = cp->gate_f;

    // everything of the first entry that does not depend on the pose is requested first: it is in flight while the
    // previous iteration is closed below
    int e0 = wave * nb_act + b;
    int2 chunk = make_int2(0, 0);
    if (e0 < n_waves) { chunk.x = tb[e0].x; chunk.y = tb[e0].y; }
    chunk.x = __builtin_amdgcn_readfirstlane(chunk.x);       // wave-uniform: scalar registers
    chunk.y = __builtin_amdgcn_readfirstlane(chunk.y);
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    v4f cert = { 0, 0, 0, 0 }, plane0 = { NAN, 0, 0, 0 };
    const bool valid0 = lane < chunk.y && chunk.x + lane < nq;
    if (valid0) {
        const int i = chunk.x + lane;
        px = G(cp->qx)[i]; py = G(cp->qy)[i]; pz = G(cp->qz)[i];                      // pointOri (:1085)
        cert = G((const v4f*)cp->cert)[i];
        plane0 = G((const v4f*)cp->plane_cache)[i];        // good as it is if the whole entry turns out to be certified
    }

    // transPointAssociateToMap (:1069-1072) and the LM trig (:1170-1175). Launch 0 of a scan gets them
    // from the host (libm); later launches rebuild them from the pose: lanes 0..2 take one angle each
    // (glibc's sinf / cosf arithmetic, see glibc_sincosf).  With solve_prev the pose is first advanced by closing
    // iteration launch-1 (LMOptimization's solve and update) right here, in every workgroup.  (The search kernel never
    // closes: the certify kernel of the same launch or k_finalize has stored the pose of this launch.)
    Fragile frag;
#pragma unroll
    for (int k = 0; k < 6; k++) frag.fr[k] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
    float T[12], sc6[6];
    if (!solve_prev && st->T_valid) {
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = st->T[k];
#pragma unroll
        for (int k = 0; k < 6; k++) sc6[k] = st->sc[k];
    } else {
        float pose[6];
        if (solve_prev) {
            static_assert(MODE == kCertify || sizeof(LmShared) <= sizeof(v4f) * kTilePts * NW, "LM scratch must fit the tile area");
            LmShared& sh = (MODE == kCertify) ? s_lm : *reinterpret_cast<LmShared*>(&s_pts[0][0]);
            const int degen0 = st->isDegenerate;
            float pose0[6];
#pragma unroll
            for (int k = 0; k < 6; k++) pose0[k] = st->pose2[(launch - 1) & 1][k];
            if (HOOK) lm_stamps[5] = wall_clock64();
            // lanes whose certificate leaves (next to) no slack fetch what their re-measurement needs behind the close
            auto early = [&]() {
                if (!HOOK && valid0 && !(cert.w > kFragileSlack) && !(ablate & 3)) {
                    const int i = chunk.x + lane;
                    frag.on = true;
                    frag.aux = G((const v4i*)cp->aux)[i];
#pragma unroll
                    for (int j = 0; j < 5; j++) frag.opos[j] = G(cp->npos)[(size_t)j * nq + i];
#pragma unroll
                    for (int j = 0; j < 6; j++) frag.fr[j] = G((const v4f*)cp->front)[(size_t)j * nq + i];
#pragma unroll
                    for (int j = 0; j < 5; j++) frag.alt[j] = G(cp->npos_alt)[(size_t)j * nq + i];
                    frag.alt_plane = G((const v4f*)cp->plane_alt)[i];
                }
            };
            auto late = [&]() {};
            const bool ended = lm_close_iteration<NW * 64, false>(cp, st, nb_act, launch - 1, blockIdx.x == 0, false, pose0, degen0,
                                                                 sh, s_lm_out, pose, HOOK ? lm_stamps : nullptr, early, late);
            if (ended) return;
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) pose[k] = st->pose2[launch & 1][k];
        }
        build_transform(pose, lane, T, sc6);
    }
    // wave-uniform by construction: into scalar registers, out of the way of everything below
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(T[k])));
#pragma unroll
    for (int k = 0; k < 6; k++) sc6[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc6[k])));
    if (HOOK) lm_stamps[6] = wall_clock64();

    WaveProf prof;
    double acc[kAcc];                                       // zeroed only after the association: nothing of pass 2 is live during pass 1
    bool defer = false;                                     // kCertify: this wave holds a lane that needs the search kernel

    if (MODE == kSearch && CNW != NW) {
        // ---- the search kernel on a partition of wider workgroups: the waves of row b in rounds of NW; per wave the association of
        // all its entries, then the residuals (on its own partition it runs the fused kernel's path below)
        for (int r = 0; r < CNW / NW; r++) {
            const int w = r * NW + wave;
            e0 = w * nb_act + b;
            for (int e = e0; e < n_waves; e += estride) {
                if (r != 0 || e != e0) {
                    chunk = make_int2(__builtin_amdgcn_readfirstlane(tb[e].x), __builtin_amdgcn_readfirstlane(tb[e].y));
                    px = 0.0f; py = 0.0f; pz = 0.0f; cert = v4f{ 0, 0, 0, 0 };
                    if (lane < chunk.y && chunk.x + lane < nq) {
                        const int i = chunk.x + lane;
                        px = G(cp->qx)[i]; py = G(cp->qy)[i]; pz = G(cp->qz)[i];
                        cert = G((const v4f*)cp->cert)[i];
                    }
                }
                associate_chunk<HOOK, (MINW > 2), !(NW == 8 && MINW == 4)>(cp, g, map, cell_start, T, gatef, ablate, chunk, lane, s_pts[wave], s_rows[wave],
                                      nullptr, px, py, pz, cert, prof);
            }
            if (HOOK) clk1 = wall_clock64();
#pragma unroll
            for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
            for (int e = e0; e < n_waves; e += estride)
                linearise_chunk<HOOK>(cp, T, sc6, make_int2(tb[e].x, tb[e].y), lane, acc);
            wave_reduce_acc(acc, lane, red[w]);
        }
        if (HOOK) clk2 = wall_clock64();
    } else {
    const bool single = e0 + estride >= n_waves;            // one entry for this wave (every scan up to 262 144 points): its point stays in registers
    if (__builtin_expect(single, 1)) {
        if (e0 < n_waves) {
            // the certificate test of associate_chunk, up front: a fully certified entry goes straight to its residuals
            float sx, sy, sz, eps;
            const bool need = lane_needs(T, px, py, pz, cert, valid0, ablate, sx, sy, sz, eps);
            // Lanes without slack (a near-tie between two neighbour distances) fail that test in every launch, however small the
            // step, and the wave that holds one would go through the association - a long way off in the instruction stream -
            // while every other wave of the launch is done.  What such a lane needs has been fetched behind the close (frag):
            // its six front members are measured right here.  If they prove the five nearest (tier B's argument against the
            // unchanged reference position: gated, the 5th nearer than anything outside the six can be, no equal distances),
            // the lane is settled on this path: tuple in the stored order - nothing to do; order changed - the plane is fitted
            // again and tuple, plane and a certificate without slack are stored.  Any lane that cannot be settled this way
            // sends the wave through the association as before.
            bool quick = false;
            if (!HOOK && __ballot(need) != 0ull) {
                bool okl = true, flip = false;
                const int nfr = min(max(frag.aux.z, 0), kNbr);                          // members of the lane's neighbourhood
                Top6k t;
#pragma unroll
                for (int k = 0; k < 6; k++) t.key[k] = kKeyInf;
                if (need) {
                    okl = frag.on && (frag.aux.y & 12) == 12 && (frag.aux.y & 3) != 0;    // tuple complete, neighbourhood valid, plane known
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        float d2;
                        const uint64_t key = make_key(frag.fr[k], sx, sy, sz, d2);       // L2_Simple order; low word: the member's map position
                        t.key[k] = (okl && k < nfr) ? key : kKeyInf;
                    }
                    sort6k(t);
                    const float r = __int_as_float(frag.aux.w) - eps;                   // everything outside the six is at least this far away
                    const float d2_5 = __uint_as_float(key_hi(t.key[4]));
                    okl = okl && key_hi(t.key[4]) < 0x7f800000u && ((double)d2_5 < cp->gate_sq) && (sqrtf(d2_5) + kCertMargin < r);
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        okl = okl && key_hi(t.key[k]) != key_hi(t.key[k + 1]);
                        flip = flip || (int)key_lo(t.key[k]) != frag.opos[k];
                    }
                }
                quick = __ballot(need && !okl) == 0ull;
                flip = flip && need && quick;
                // A flip back to the tuple the lane had before (two nearly equidistant neighbours swapping places with every
                // micro-step) finds that tuple's plane kept: the two are exchanged.  Anything else is fitted again (:1099-1122),
                // and the tuple it replaces is kept with its plane.
                bool back = flip && (frag.aux.y & 16) != 0;
#pragma unroll
                for (int k = 0; k < 5; k++) back = back && (int)key_lo(t.key[k]) == frag.alt[k];
                v4f pl_new = frag.alt_plane;
                if (__ballot(flip && !back)) {
                    if (flip && !back) {
                        float qr[5][3], mx[5][3];
#pragma unroll
                        for (int j = 0; j < 5; j++) {                                   // the five nearest, in order (:1099-1101)
                            float x = 0.0f, y = 0.0f, z = 0.0f;
#pragma unroll
                            for (int k = 0; k < 6; k++) {
                                const bool is = k < nfr && __float_as_int(frag.fr[k].w) == (int)key_lo(t.key[j]);   // (rows beyond the member count hold stale positions)
                                x = is ? frag.fr[k].x : x; y = is ? frag.fr[k].y : y; z = is ? frag.fr[k].z : z;
                            }
                            qr[j][0] = x; qr[j][1] = y; qr[j][2] = z; mx[j][0] = x; mx[j][1] = y; mx[j][2] = z;
                        }
                        float X[3];
                        plane_fit_5x3(qr, X);                                           // :1104
                        float pa = X[0], pb = X[1], pc = X[2], pd = 1.0f;
                        const float ps = sqrtf(pa * pa + pb * pb + pc * pc);             // :1111
                        pa /= ps; pb /= ps; pc /= ps; pd /= ps;
                        bool planeValid = true;
#pragma unroll
                        for (int j = 0; j < 5; j++) {                                   // :1115-1122
                            const float rr_ = pa * mx[j][0] + pb * mx[j][1] + pc * mx[j][2] + pd;
                            if ((double)fabsf(rr_) > cp->plane_tol) planeValid = false;
                        }
                        pl_new = v4f{ planeValid ? pa : NAN, pb, pc, pd };
                    }
                }
                if (flip) {
                    const int i = chunk.x + lane;
                    G((v4f*)cp->plane_alt)[i] = plane0;                                 // what the lane had until now
#pragma unroll
                    for (int k = 0; k < 5; k++) G(cp->npos_alt)[(size_t)k * nq + i] = frag.opos[k];
                    G((v4f*)cp->plane_cache)[i] = pl_new;
#pragma unroll
                    for (int k = 0; k < 5; k++) G(cp->npos)[(size_t)k * nq + i] = (int)key_lo(t.key[k]);
                    plane0 = pl_new;
                    // the stored certificate spoke of the old order: none from now on (the reference position and the radii the
                    // neighbourhood vouches for stay as they are)
                    const v4f cnew = { cert.x, cert.y, cert.z, 0.0f };
                    G((v4f*)cp->cert)[i] = cnew;
                    const v4i anew = { frag.aux.x, (frag.aux.y & ~3) | ((pl_new.x == pl_new.x) ? 1 : 2) | 16, frag.aux.z, frag.aux.w };
                    G((v4i*)cp->aux)[i] = anew;
                }
            }
            if (MODE == kCertify) {
                defer = __ballot(need) != 0ull && !quick;                            // this workgroup's row is the search kernel's business
            } else if (HOOK || __builtin_expect(__ballot(need) != 0ull && !quick, 0)) {     // (unlikely: the search is laid out away from the certified path)
                // Point and plane are not kept in registers through the association: what the certified path holds in registers
                // must not be live across the search, or the allocator spills it on the common path (measured: +1.1 us on every
                // steady launch).  In the builds with one 8-wave workgroup per CU they are parked in LDS (the association writes the planes it fits there
                // too): reading them back from memory queues the loads behind the association's fourteen stores.
                float* lkeep = kKeep ? &s_keep[kKeep ? wave : 0][0][0] : nullptr;
                if (kKeep) {
                    lkeep[0 * 64 + lane] = px; lkeep[1 * 64 + lane] = py; lkeep[2 * 64 + lane] = pz;
                    lkeep[3 * 64 + lane] = plane0.x; lkeep[4 * 64 + lane] = plane0.y; lkeep[5 * 64 + lane] = plane0.z; lkeep[6 * 64 + lane] = plane0.w;
                }
                associate_chunk<HOOK, (MINW > 2), !(NW == 8 && MINW == 4)>(cp, g, map, cell_start, T, gatef, ablate, chunk, lane, s_pts[wave], s_rows[wave],
                                      lkeep, px, py, pz, cert, prof);
                asm volatile("" ::: "memory");
                if (kKeep) {
                    wave_lds_sync();
                    px = lkeep[0 * 64 + lane]; py = lkeep[1 * 64 + lane]; pz = lkeep[2 * 64 + lane];
                    plane0 = v4f{ lkeep[3 * 64 + lane], lkeep[4 * 64 + lane], lkeep[5 * 64 + lane], lkeep[6 * 64 + lane] };
                } else if (valid0) {
                    const int i = chunk.x + lane;
                    px = G(cp->qx)[i]; py = G(cp->qy)[i]; pz = G(cp->qz)[i];
                    plane0 = G((const v4f*)cp->plane_cache)[i];
                }
            }
            if (HOOK) clk1 = wall_clock64();
#pragma unroll
            for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
            if (valid0 && !defer) linearise_point<HOOK>(cp, T, sc6, chunk.x + lane, px, py, pz, plane0, acc);
        } else {
#pragma unroll
            for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
        }
    } else {
    // A wave with several entries first asks whether any of its points needs more than its certificate: in the steady state
    // of the loop none does, and the wave goes straight to the residuals (nothing of the association is live there).
    bool any_need = HOOK;
    if (!HOOK) {
        for (int e = e0; e < n_waves; e += estride) {
            const int2 en = make_int2(__builtin_amdgcn_readfirstlane(tb[e].x), __builtin_amdgcn_readfirstlane(tb[e].y));
            const int i = en.x + lane;
            bool need = false;
            if (lane < en.y && i < nq) {
                const float qx = G(cp->qx)[i], qy = G(cp->qy)[i], qz = G(cp->qz)[i];
                const v4f ce = G((const v4f*)cp->cert)[i];
                float sx, sy, sz, eps;
                need = lane_needs(T, qx, qy, qz, ce, true, ablate, sx, sy, sz, eps);
            }
            any_need = any_need || (__ballot(need) != 0ull);
        }
    }
    if (MODE == kCertify) defer = any_need;
    else {
    // ---- pass 1: associate
    if (__builtin_expect(any_need, 0))
    for (int e = e0; e < n_waves; e += estride) {
        if (e != e0) {
            chunk = make_int2(__builtin_amdgcn_readfirstlane(tb[e].x), __builtin_amdgcn_readfirstlane(tb[e].y));
            px = 0.0f; py = 0.0f; pz = 0.0f; cert = v4f{ 0, 0, 0, 0 };
            if (lane < chunk.y && chunk.x + lane < nq) {
                const int i = chunk.x + lane;
                px = G(cp->qx)[i]; py = G(cp->qy)[i]; pz = G(cp->qz)[i];
                cert = G((const v4f*)cp->cert)[i];
            }
        }
        associate_chunk<HOOK, (MINW > 2), !(NW == 8 && MINW == 4)>(cp, g, map, cell_start, T, gatef, ablate, chunk, lane, s_pts[wave], s_rows[wave],
                              nullptr, px, py, pz, cert, prof);
    }
    }
    if (HOOK) clk1 = wall_clock64();

    // ---- pass 2: linearise
#pragma unroll
    for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
    if (!defer)
    for (int e = e0; e < n_waves; e += estride)
        linearise_chunk<HOOK>(cp, T, sc6, make_int2(tb[e].x, tb[e].y), lane, acc);
    }
    if (HOOK) clk2 = wall_clock64();
    wave_reduce_acc(acc, lane, red[wave]);
    }

    if (HOOK && cp->dbg_clk && lane == 0) {
        const auto d = G(cp->dbg_clk) + kProfWords * ((size_t)blockIdx.x * NW + wave);
        d[0] = tk_start; d[1] = clk1; d[2] = clk2; d[3] = wall_clock64();
        d[4] = (unsigned long long)prof.mode; d[5] = (unsigned long long)prof.rows; d[6] = (unsigned long long)prof.pts; d[7] = (unsigned long long)prof.raw;
        d[8] = (unsigned long long)prof.n_a; d[9] = (unsigned long long)prof.n_b; d[10] = (unsigned long long)chunk.y; d[11] = (unsigned long long)prof.n_c;
        d[12] = prof.ts[0]; d[13] = (unsigned long long)prof.why; d[14] = prof.ts[1]; d[15] = prof.ts[2];
        // fused LM close (solve_prev launches): entry, partial sums reduced, normal equations, QR, update, barrier, T built
        d[16] = lm_stamps[5]; d[17] = lm_stamps[0]; d[18] = lm_stamps[1]; d[19] = lm_stamps[2]; d[20] = lm_stamps[3]; d[21] = lm_stamps[4]; d[22] = lm_stamps[6]; d[23] = prof.ts[3];
        d[24] = (unsigned long long)prof.n_fb; d[25] = (unsigned long long)prof.reach_mm; d[26] = (unsigned long long)prof.kq; d[27] = (unsigned long long)prof.cmax;
    }
    if (MODE == kCertify) {
        // one lane that needs the search kernel sends the whole workgroup there: its row has to come out of one summation
        if (__syncthreads_or(defer ? 1 : 0)) {
            if (tid == 0) {
                const int slot = atomicAdd(cp->wl_count + (launch & 1), 1);
                G(cp->wl_items)[(size_t)(launch & 1) * (size_t)nblocks + slot] = b;
            }
            return;
        }
    } else
        __syncthreads();
    write_partial_row<CNW>(cp, launch, b, tid, red);
    if (!listed) return;
    wl_at += (int)gridDim.x;
    __syncthreads();                                      // (the next item reuses the workgroup's LDS)
    }
    // ---- kFlagCloseAfter: this workgroup is the only one of its slot in this launch and every row of the iteration is written
    // (by the certify kernel before this launch, or above): what k_finalize would do in a launch of its own
    if (MODE == kSearch) {
        __syncthreads();
        LmShared& sh = *reinterpret_cast<LmShared*>(&s_pts[0][0]);
        static_assert(MODE != kSearch || sizeof(LmShared) <= sizeof(v4f) * kTilePts * NW, "LM scratch must fit the tile area");
        float pose0[6], pose[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pose0[k] = st->pose2[launch & 1][k];
        const int degen0 = st->isDegenerate;
        if (tid == 0) { cp->wl_count[0] = 0; cp->wl_count[1] = 0; }
        const bool ended = lm_close_iteration<NW * 64, false>(cp, st, nb_act, launch, true, false, pose0, degen0, sh, s_lm_out, pose);
        if (!ended && tid < 64) {
            float Tn[12], scn[6];
            build_transform(pose, tid, Tn, scn);
            if (tid == 0 && !st->stalled) {
#pragma unroll
                for (int k = 0; k < 12; k++) st->T[k] = Tn[k];
#pragma unroll
                for (int k = 0; k < 6; k++) st->sc[k] = scn[k];
                st->T_valid = 1;
            }
        }
    }
}
