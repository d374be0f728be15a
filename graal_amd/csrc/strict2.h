// strict2.h -- reference arithmetic over the step's UNION SET (strict_sets.h): k_gprep + k_strict2.  Included by graal_hip.hip
// inside its anonymous namespace, behind the per-neighbour strict kernels whose helpers it uses.
//
// Same sums as k_strict_dense (the O(m^2) validation kernel: every pixel of contig(fA) u contig(fB_k) under every candidate,
// kernels3.cu:3259-3718 as written) -- every term it adds that is not exactly zero, each exactly once per class of equal inputs:
//   * the K sets of a step are priced TOGETHER: the current layout's value of a fragment pair is computed once per step, not once
//     per neighbour, and candidates of different neighbours that hand the model the same inputs (eject / flip of fA: all K of them;
//     the K sets are mostly the same contigs) share one evaluation (classes per pair of GLOBAL pieces: k_gprep);
//   * tiles never straddle a global piece, so the class list of a unit is WAVE-UNIFORM: the class loop is scalar, the lanes' side
//     of a class (transformed start, orientation, sub-fragment centres) is computed once per (unit, class), the segment side once
//     per (fragment, class) by one lane each; what remains per (pair, class) is the contact model itself;
//   * the lanes of a wave take the fuller tile of a tile pair (a cut fragment is a tile of its own).
// Unit list by the cull blocks of k_gprep (interval arithmetic on tile extents, as k_strict_cull), dealt to ranks by (ti + tj) % world.

struct S2Args {
    USet* uset;            // written by block 0 of k_gprep
    GClass* cls;           // [US_MAXPAIRS][US_NCAND]
    int* cls_n;            // [US_MAXPAIRS]
    int seg_unit;          // fragments of the segment side per entry of the unit list (k_gprep), a power of two
    int rep_max;           // waves that may share one unit's classes at most (1, 2, 4, 8)
    unsigned long long target;   // units the grid wants (a few per wave)
    int draw_min;                // units per wave from which the waves draw their units from the counter instead of taking every n-th
    unsigned long long* next;    // the next unit nobody has taken yet, minus the grid's waves (zero at rest: the step's last block clears it)
    // k_strict2 WITHOUT an event behind k_gprep (a grid of <= 512 blocks, one rank): k_gprep's last block stores the step's number in gp[GP_DONE] (gp[0]
    // is its blocks' ticket), k_strict2's blocks wait for it -- bounded -- before they read anything of k_gprep's.  nullptr: ordered by an event
    unsigned long long* gp;
    unsigned long long gp_seq;
    int gp_wait_ticks;
    int gp_acquire;              // 1: a block's first wave runs the agent-scope acquire also when it found the word at once
};

#if defined(GRAAL_STAMPS) && defined(GRAAL_S2_COUNTS)   // (work counters: per-lane atomics -- they distort the stamps' timeline)
#define S2_COUNT(i, v) do { atomicAdd(&g_hitstat[i], (unsigned long long)(v)); } while (0)
#else
#define S2_COUNT(i, v) do { } while (0)
#endif

constexpr int GP_DONE = 32;            // S2Args::gp: [0] k_gprep's ticket, [GP_DONE] its completion word (a line of its own: 512 blocks poll it)
constexpr int S2_BAL_MAX = 1024;       // units of a step up to which k_strict2 deals its waves to them by their classes
constexpr int GPREP_CLS_BLOCKS = 144;  // blocks of k_gprep that build classes (one wave per piece pair: 561 pairs at most, 576 waves); the rest cull

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// device-scope (write-through) stores: what k_gprep hands to a k_strict2 that is not ordered behind it by an event must be visible to the
// other XCDs without a write-back of the writer's L2 (__threadfence: a microsecond per block, one after the other) -- like the scan's queue
__device__ __forceinline__ void st_dev(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// old bp extent [lo, hi) of tile t (fragments of ONE global piece, consecutive positions of one contig)
__device__ __forceinline__ void utile_extent(const USet& U, const int* __restrict__ pstart, int t, int& lo, int& hi, int& g, int& cnt)
{
    int off;
    g = utile_piece(U, t, off);
    const UPiece& P = U.p[g];
    const UContig& C = U.c[P.contig];
    const int first = P.lo + off * U.tile;
    const int left = P.n - off * U.tile;
    cnt = left < U.tile ? left : U.tile;
    lo = pstart[C.base + first];
    hi = first + cnt >= C.len ? C.lbp : pstart[C.base + first + cnt];
}

__global__ __launch_bounds__(256) void k_gprep(const NbTables* __restrict__ tabs, const int* __restrict__ pstart, int fA, int K, int rank, int world,
                                                int reach_bp, int no_window, int quirk, int seg_unit, int tile_frags, unsigned long long* __restrict__ list,
                                                unsigned long long* __restrict__ list_n, unsigned long long cap,
                                                unsigned long long* __restrict__ counters, S2Args s2)
{
    __shared__ USet s_U;
    __shared__ UEnd s_A, s_B[MAXK];
    __shared__ PieceKey s_keys[MAXK];
    __shared__ unsigned s_live, s_mass;
    __shared__ int s_cuts[US_MAXC * (US_MAXK + 1)], s_ncut[US_MAXC];
    __shared__ Xf s_xf[MAXK][N_OPS][NP];
    __shared__ int s_ne[256], s_off[257], s_tp[256], s_pr[256];
    __shared__ unsigned long long s_key2[4][US_NCAND][2];    // the class blocks' candidate keys (two 64-bit words each), per wave
    __shared__ unsigned long long s_base;
    const int t = threadIdx.x, lane = t & 63, wib = rfl(t >> 6);
    STAMP(11, blockIdx.x == 0 && t == 0);
    STAMP_MAX(25, t == 0);
    if (t == 0) { s_live = 0; s_mass = 0; }
    __syncthreads();
    if (t <= K) {   // thread K: fA; threads 0..K-1: the neighbours.  (The ends' records come from k_tm's tables: fragment -> geometry -> contig base
                    // were three dependent round trips, 10 us of them under a scan that saturates the memory system)
        if (t == K) s_A = tabs[0].endA;
        else {
            s_B[t] = tabs[t].endB;
            s_keys[t] = tabs[t].key;
            if (tabs[t].fB != fA) atomicOr(&s_live, 1u << t);
            if (tabs[t].set_m > 0) atomicOr(&s_mass, 1u << t);   // (0: fB == fA, or its fragment pairs were priced by k_tm already)
        }
    }
    for (int i = t; i < K * N_OPS * NP; i += 256) { const int k = i / (N_OPS * NP), r = i - k * (N_OPS * NP); s_xf[k][r / NP][r % NP] = tabs[k].xf[r / NP][r % NP]; }
    __syncthreads();
    STAMP(21, blockIdx.x == 0 && t == 0);
    if (t == 0) uset_build_geometry(s_U, s_A, s_B, K, s_live, s_mass, s_cuts, s_ncut, tile_frags);
    __syncthreads();
    STAMP(22, blockIdx.x == 0 && t == 0);
    for (int i = t; i < s_U.n_pieces * US_MAXK; i += 256) uset_piece_pk(s_U, s_keys, K, i / US_MAXK, i % US_MAXK);
    __syncthreads();
    const USet& U = s_U;
    if (blockIdx.x == 0) {
        int* dst = reinterpret_cast<int*>(s2.uset);
        const int* src = reinterpret_cast<const int*>(&s_U);
        for (int i = t; i < (int)(sizeof(USet) / 4); i += 256) st_dev(&dst[i], src[i]);
    }
    const int np = U.n_pieces;
    STAMP(12, blockIdx.x == 0 && t == 0);
    if ((int)blockIdx.x < GPREP_CLS_BLOCKS) {
        // ---- classes of equal inputs per pair of global pieces: one wave per pair.  Lane l holds the candidates l, l + 64, l + 128
        // (candidate = k * 13 + op); a class is the ballot of the lanes whose key equals the first unassigned candidate's.
        const int n_pairs = np * (np + 1) / 2;
        for (int pi = blockIdx.x * 4 + wib; pi < n_pairs; pi += GPREP_CLS_BLOCKS * 4) {
            int g = 0, rem = pi;
            while (rem >= np - g) { rem -= np - g; g++; }
            const int h = g + rem;
            const int pair = upair_index(g, h);
            const InputsKey old = inputs_key(uxf_old(U, g), uxf_old(U, h), quirk != 0);
            InputsKey key[3];
            Xf xa[3], xb[3];
            bool un[3], ms[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int cand = lane + 64 * r, k = cand / N_OPS, op = cand - k * N_OPS;
                un[r] = false;
                ms[r] = cand < K * N_OPS && ((U.mass >> k) & 1u);
                key[r] = old;
                xa[r] = uxf_old(U, g); xb[r] = xa[r];
                if (cand < K * N_OPS) {
                    const int pg = U.p[g].pk[k], ph = U.p[h].pk[k];
                    if (pg && ph) {
                        xa[r] = s_xf[k][op][pg]; xb[r] = s_xf[k][op][ph];
                        key[r] = inputs_key(xa[r], xb[r], quirk != 0);
                        un[r] = !ikey_eq(key[r], old);
                    }
                }
            }
            GClass* out = s2.cls + (size_t)pair * US_NCAND;
            int nc = 0;
            STAMP_MAX(26, lane == 0);
            // Classes without rounds: every candidate compares its key with every candidate that differs from the current layout (their keys go
            // round through LDS, one wave-uniform broadcast read each) and so collects the 130-bit mask of ITS class; the class's lowest
            // candidate leads it, a leader's number is the count of leaders below it, and it writes the record from the transforms it already
            // holds.  (One round per class -- first unassigned candidate, ballots, its key, compares -- was a chain of 0.2 us per class in a wave
            // that has its SIMD to itself: 6-10 us of this kernel per pair; this is 0.6-3 us whatever the number of classes.)
            unsigned long long (*const kk)[2] = s_key2[wib];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int cand = lane + 64 * r;
                if (cand < K * N_OPS) {
                    kk[cand][0] = (unsigned long long)(unsigned)key[r].x | ((unsigned long long)(unsigned)key[r].y << 32);
                    kk[cand][1] = (unsigned long long)(unsigned)key[r].z | ((unsigned long long)(unsigned)key[r].w << 32);
                }
            }
            unsigned long long ka[3], kb[3], mm0[3] = {0ull, 0ull, 0ull}, mm1[3] = {0ull, 0ull, 0ull};
            unsigned mm2[3] = {0u, 0u, 0u};
#pragma unroll
            for (int r = 0; r < 3; r++) {
                ka[r] = (unsigned long long)(unsigned)key[r].x | ((unsigned long long)(unsigned)key[r].y << 32);
                kb[r] = (unsigned long long)(unsigned)key[r].z | ((unsigned long long)(unsigned)key[r].w << 32);
            }
            const unsigned long long u0 = __ballot(un[0]), u1 = __ballot(un[1]), u2 = __ballot(un[2]);
            const int nr = (K * N_OPS + 63) >> 6;      // candidates per lane that exist at this K (wave-uniform)
            WAVE_LDS_SYNC();
            for (unsigned long long bits = u0; bits; bits &= bits - 1ull) {
                const int j = __ffsll((long long)bits) - 1;
                const unsigned long long ja = kk[j][0], jb = kk[j][1];
                const unsigned long long bit = 1ull << j;
                mm0[0] |= (ka[0] == ja && kb[0] == jb) ? bit : 0ull;
                if (nr > 1) mm0[1] |= (ka[1] == ja && kb[1] == jb) ? bit : 0ull;
                if (nr > 2) mm0[2] |= (ka[2] == ja && kb[2] == jb) ? bit : 0ull;
            }
            for (unsigned long long bits = u1; bits; bits &= bits - 1ull) {
                const int j = __ffsll((long long)bits) - 1;
                const unsigned long long ja = kk[64 + j][0], jb = kk[64 + j][1];
                const unsigned long long bit = 1ull << j;
                mm1[0] |= (ka[0] == ja && kb[0] == jb) ? bit : 0ull;
                mm1[1] |= (ka[1] == ja && kb[1] == jb) ? bit : 0ull;
                if (nr > 2) mm1[2] |= (ka[2] == ja && kb[2] == jb) ? bit : 0ull;
            }
            for (unsigned bits = (unsigned)(u2 & 3ull); bits; bits &= bits - 1u) {
                const int j = __ffs((int)bits) - 1;
                const unsigned long long ja = kk[128 + j][0], jb = kk[128 + j][1];
                const unsigned bit = 1u << j;
                mm2[0] |= (ka[0] == ja && kb[0] == jb) ? bit : 0u;
                mm2[1] |= (ka[1] == ja && kb[1] == jb) ? bit : 0u;
                mm2[2] |= (ka[2] == ja && kb[2] == jb) ? bit : 0u;
            }
            // (a candidate whose key is the current layout's belongs to no class: it is compared with nobody -- the loops run over the others --
            // and leads nothing)
            bool lead[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int lowest = mm0[r] ? __ffsll((long long)mm0[r]) - 1 : (mm1[r] ? 64 + __ffsll((long long)mm1[r]) - 1 : (mm2[r] ? 128 + __ffs((int)mm2[r]) - 1 : -1));
                lead[r] = un[r] && lowest == lane + 64 * r;
            }
            const unsigned long long l0 = __ballot(lead[0]), l1 = __ballot(lead[1]), l2 = __ballot(lead[2]);
            const unsigned long long below = (1ull << lane) - 1ull;
            const int n0 = __popcll(l0), n1 = __popcll(l1);
            nc = n0 + n1 + __popcll(l2);
            const unsigned long long ms0 = __ballot(ms[0]), ms1 = __ballot(ms[1]), ms2 = __ballot(ms[2]);
#pragma unroll
            for (int r = 0; r < 3; r++)
                if (lead[r]) {
                    const int idx = (r == 0 ? 0 : (r == 1 ? n0 : n0 + n1)) + __popcll((r == 0 ? l0 : (r == 1 ? l1 : l2)) & below);
                    GClass c = gclass_make(xa[r], xb[r], lane + 64 * r);
                    c.m0 = mm0[r]; c.m1 = mm1[r]; c.m2 = mm2[r] & 3u;
                    c.w0 = mm0[r] & ms0; c.w1 = mm1[r] & ms1; c.w2 = mm2[r] & (unsigned)(ms2 & 3ull);
                    {   // (eight 8-byte device-scope stores)
                        static_assert(sizeof(GClass) == 64, "GClass is written as eight 64-bit words");
                        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&c);
                        unsigned long long* o = reinterpret_cast<unsigned long long*>(&out[idx]);
#pragma unroll
                        for (int q = 0; q < 8; q++) st_dev(&o[q], w[q]);
                    }
                }
            WAVE_LDS_SYNC();   // (the next pair of this wave writes its keys over these)
            if (lane == 0) st_dev(&s2.cls_n[pair], nc);
        }
        STAMP_MAX(13, lane == 0);
    } else {
    // ---- unit list: rows = tiles; a tile pair is listed iff some fragment pair of it can be inside the window under the current
    // layout or under some candidate whose inputs differ from the current layout's (interval arithmetic, no per-pair work)
    const int n_tiles = U.n_tiles;
    // (entries of seg_unit fragments of the segment side: k_strict2 merges neighbouring entries or deals one to several waves, by their number)
    const int seg = seg_unit, R = 1;
    const int n_cull = (int)gridDim.x - GPREP_CLS_BLOCKS;
    for (int ti = (int)blockIdx.x - GPREP_CLS_BLOCKS; ti < n_tiles; ti += n_cull) {
        // (the row's extent by every thread for itself, next to its own tile's: the same two addresses for all lanes, one round trip for both tiles --
        // fetched by thread 0 and handed round through LDS they were two more dependent round trips in front of every row)
        int xlo, xhi, g, cnt_i;
        utile_extent(U, pstart, ti, xlo, xhi, g, cnt_i);
        const int cx = U.p[g].contig;
        for (int tj0 = ti; tj0 < n_tiles; tj0 += 256) {
            const int tj = tj0 + t;
            bool alive = false;
            int cnt_j = 0, h = g;
            if (tj < n_tiles && ((ti + tj) % world) == rank) {
                int ylo, yhi;
                utile_extent(U, pstart, tj, ylo, yhi, h, cnt_j);
                const bool near_old = U.p[h].contig == cx && max(ylo - xhi, xlo - yhi) <= reach_bp;
                const InputsKey old = inputs_key(uxf_old(U, g), uxf_old(U, h), quirk != 0);
                for (int k = 0; k < K && !alive; k++) {
                    const int pg = U.p[g].pk[k], ph = U.p[h].pk[k];
                    if (!pg || !ph || !((U.mass >> k) & 1u)) continue;
                    for (int op = 0; op < N_OPS; op++) {
                        const Xf a = s_xf[k][op][pg], b = s_xf[k][op][ph];
                        if (ikey_eq(inputs_key(a, b, quirk != 0), old)) continue;   // the current layout's inputs: nothing to price
                        if (no_window || near_old) { alive = true; break; }
                        if (a.label != b.label) continue;
                        const int xs2 = a.sigma > 0 ? xlo + a.off : a.off - xhi, xe2 = a.sigma > 0 ? xhi + a.off : a.off - xlo;
                        const int ys2 = b.sigma > 0 ? ylo + b.off : b.off - yhi, ye2 = b.sigma > 0 ? yhi + b.off : b.off - ylo;
                        if (max(ys2 - xe2, xs2 - ye2) <= reach_bp) { alive = true; break; }
                    }
                }
            }
            const int lanes_first = cnt_i >= cnt_j ? 1 : 0;
            const int cs = lanes_first ? cnt_j : cnt_i;
            const int ne = alive ? ((cs + seg - 1) / seg) * R : 0;
            // the block writes the units of its alive tile pairs together: thread o takes entry o of the round (a tile pair has up to 64 / seg x R
            // of them: written by its own thread, one after the other, they were most of this kernel at mid-size shapes)
            s_ne[t] = ne;
            s_tp[t] = tj | (lanes_first << 16) | (cs << 17);
            s_pr[t] = upair_index(g, h);
            __syncthreads();
            if (t < 64) wave_excl_scan(s_ne, s_off, 256);
            __syncthreads();
            const int total = s_off[256];
            if (total > 0) {
                if (t == 0) s_base = atomicAdd(list_n, (unsigned long long)total);
                __syncthreads();
                const unsigned long long base = s_base;
                for (int o = t; o < total; o += 256) {
                    int lo = 0, hi = 255;     // last thread whose first entry is <= o
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= o) lo = mid; else hi = mid - 1; }
                    const int e = o - s_off[lo], tp = s_tp[lo];
                    const int cs_o = (tp >> 17) & 127, j0 = (e / R) * seg, c = cs_o - j0 < seg ? cs_o - j0 : seg;
                    const unsigned long long at = base + (unsigned long long)o;
                    if (at < cap) st_dev(&list[at], uunit_pack(ti, tp & 0xffff, j0, c, (tp >> 16) & 1, e % R, R, s_pr[lo]));
                    else atomicOr(&counters[6], 2ull);   // (cannot happen: the host sizes the list)
                }
            }
            __syncthreads();
        }
    }
    STAMP_MAX(14, t == 0);
    }
    if (s2.gp != nullptr) {
        // "k_gprep is complete": every block waits for its own device-scope stores (and the list's atomic) to be acknowledged and takes a ticket;
        // the last one puts the ticket back and stores the step's number
        ATOMICS_DONE();
        __syncthreads();
        if (t == 0) {
            const unsigned long long ticket = atomicAdd(&s2.gp[0], 1ull);
            if (ticket == (unsigned long long)gridDim.x - 1ull) {
                st_dev(&s2.gp[0], 0ull);
                // (relaxed: a release here is a write-back of this XCD's whole L2 -- 15 us with the step's dirty lines in it -- and nothing needs it: every
                // block's results are device-scope stores, acknowledged before its ticket; the last ticket is taken behind all of them)
                st_dev(&s2.gp[GP_DONE], s2.gp_seq);
                STAMP(24, true);
            }
        }
    }
}

struct STile2 { int start_bp, len_bp, flags, frag; Stat st; };   // one staged fragment of the segment side, 48 bytes

// one slot pair in one layout.  c_first / c_second: centres (kb) of the two sub-fragments, first = the fragment that comes first in the
// union's order (the `X` of ex_pair); the same operations in the same order as ex_pair / ex_pair_ref
__device__ __forceinline__ float s2_ex(bool cis, float c_first, float c_second, float norm, int circ, float s_tot, const Par& p)
{
    if (!cis) return p.v_inter * norm;
    const float s = fabsf(c_second - c_first);
    if (circ) {   // (rare, wave-uniform.  The barrier keeps the circular model's arithmetic inside the branch: the compiler used to pair its
                  // divisions with the linear model's -- packed float32 -- and computed them for every pair)
        float s2 = s;
        asm volatile("" : "+v"(s2));
        return rippe_circ(s2, s_tot, p) * norm;
    }
    return rippe(s, p) * norm;
}

// candidates of a class (bits of m0, m1, m2) += v, in the block's LDS sums: lane l takes the candidates l, l + 64, l + 128
__device__ __forceinline__ void s2_add_mask(long long* __restrict__ acc, unsigned long long m0, unsigned long long m1, unsigned m2, long long v, int lane)
{
    if ((m0 >> lane) & 1ull) atomicAdd((unsigned long long*)&acc[lane], (unsigned long long)v);
    if ((m1 >> lane) & 1ull) atomicAdd((unsigned long long*)&acc[64 + lane], (unsigned long long)v);
    if (lane < 2 && ((m2 >> lane) & 1u)) atomicAdd((unsigned long long*)&acc[128 + lane], (unsigned long long)v);
}
// a term beyond Q30's range (to_coarse): into the candidates' coarse sums, or their flags -- by the lane that met it (rare)
__device__ __forceinline__ void s2_coarse_mask(long long* __restrict__ acc, unsigned long long* __restrict__ nf, unsigned long long m0, unsigned long long m1,
                                               unsigned m2, double v)
{
    long long c;
    const bool ok = to_coarse(v, c);
    unsigned long long mm[3] = {m0, m1, (unsigned long long)m2};
    for (int r = 0; r < 3; r++) {
        unsigned long long m = mm[r];
        while (m) {
            const int cand = 64 * r + __ffsll((long long)m) - 1;
            m &= m - 1ull;
            if (ok) atomicAdd((unsigned long long*)&acc[MAXK * N_OPS + cand], (unsigned long long)c);
            else nf_flag(nf, cand / N_OPS, cand % N_OPS);
        }
    }
}
__device__ __forceinline__ void s2_flag_mask(unsigned long long* nf, unsigned long long m0, unsigned long long m1, unsigned m2, int lane)
{
    for (int r = 0; r < 3; r++) {
        const int cand = lane + 64 * r;
        const bool in = r == 0 ? ((m0 >> lane) & 1ull) : (r == 1 ? ((m1 >> lane) & 1ull) : (lane < 2 && ((m2 >> lane) & 1u)));
        if (in) nf_flag(nf, cand / N_OPS, cand % N_OPS);
    }
}

template <bool MULTI>
__global__ __launch_bounds__(256, 4) void k_strict2(FinArgs fa, StrictArgs sa, S2Args s2, int K, const unsigned long long* __restrict__ list,
                                                  unsigned long long* __restrict__ list_n, long long* __restrict__ d_q_out,
                                                  volatile long long* host_res, long long seq)
{
    constexpr int SEG = MULTI ? 4 : 16;        // fragments of a unit's segment at most (several sub-fragments: 9 slot pairs per fragment pair
                                               // already amortise a class's set-up, and the current layout's values of a unit live in LDS: two
                                               // fragments of the segment per lane -- with tiles of 32 fragments the two HALVES of the wave take two
                                               // fragments of the segment at a time, so a unit is 32 x 4 instead of 64 x 2: the same work in a squarer
                                               // block, which wastes fewer lanes on pieces of a few dozen bins and at the window's edge)
    constexpr int EXO_SEG = MULTI ? 2 : 16;    // ... segment fragments PER LANE whose current-layout values are kept
    constexpr int NS = MULTI ? 3 : 1;          // sub-fragment slots per bin at most
    constexpr int NSP = NS * NS;
    const Geo* __restrict__ geo = fa.geo;
    const Stat* __restrict__ stat = fa.stat;
    unsigned long long* __restrict__ counters = fa.counters;
    // (wib is NOT declared wave-uniform to the compiler: with readfirstlane on it the kernel needs no scratch memory -- 60-72 bytes of it are spilled
    // outside the pair loops otherwise -- but the pair loops come out 3 % slower: C5's 7 contigs 1.41 ms against 1.36, tools/ab_late.sh)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    __shared__ USet s_U;
    __shared__ long long s_acc[US_NCAND];
    __shared__ STile2 s_tile[4][SEG];
    __shared__ float s_cy[4][SEG][4];          // the segment side's centres (and its orientation, [3]) in the layout being priced
    __shared__ float s_exo[4][EXO_SEG * NSP][64];  // the current layout's values of the unit's slot pairs, [segment fragment of the lane][slot pair][lane]
    __shared__ int s_last;
    constexpr int CLS_CHUNK = 32;              // class records of a unit's piece pair staged per wave (a pair has up to 130; most have a dozen)
    __shared__ GClass s_cls[4][CLS_CHUNK];
    __shared__ unsigned long long s_ent[4][16];   // the unit-list entries a wave is working through
    __shared__ int s_pref[S2_BAL_MAX + 1], s_bsum[256], s_boff[257];   // the waves of a step's few units (see `bal`)
#if defined(GRAAL_STAMPS) && defined(GRAAL_S2_COUNTS)
    __shared__ float s_sold[4][SEG][64];       // (counting build: the current layout's float32 distance of every pair of the unit)
#endif
    STAMP(16, blockIdx.x == 0 && threadIdx.x == 0);
    int skip = fa.skip;
    if (s2.gp != nullptr) {
        // no event in front of this kernel: k_gprep (on the other stream; launched first, and this grid leaves room for its blocks on every CU)
        // says when the union set, the classes and the unit list are complete.  Bounded: if the two kernels do not run side by side -- a tool
        // that serialises dispatches -- the step ends as failed and the host repeats it behind an event (eval_sync: spin_used)
        if (threadIdx.x == 0) {
            const unsigned long long t_end = wall_clock64() + (unsigned long long)s2.gp_wait_ticks;
            bool ok = false, waited = false;
            for (;;) {
                if (__hip_atomic_load(&s2.gp[GP_DONE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s2.gp_seq) { ok = true; break; }
                if (wall_clock64() > t_end) break;
                waited = true;
                __builtin_amdgcn_s_sleep(16);   // (~0.4 us between polls: 512 blocks poll one word)
            }
            if (!ok) atomicOr(&fa.counters[6], 8ull);
            s_last = ok ? 0 : 1;
            // The consumer side of the hand-off as the memory model asks for it: ONE relaxed poll, ONE agent-scope acquire by ONE wave of the block
            // (the caches are the CU's and the XCD's, not the wave's: all 2,048 waves of the grid doing it were 15 us of this kernel's prologue),
            // the wait for the invalidate, the barrier, then plain loads.  Round 4 skipped the acquire in a block that found the word at once
            // (nothing of k_gprep's is read by this grid before the word, and the kernel's own start invalidated what earlier steps left: an
            // argument from cache behaviour, not from the model; GRAAL_GP_ACQUIRE=0 keeps that form for A/B -- 21,720 steps x 39 sums at the C2
            // stand-in were bit-identical either way, profiles/r05_handoff_soak.log, and the step costs the same, 117-127 us)
            if (waited || s2.gp_acquire) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the invalidate completes asynchronously: the barrier below must not release the other waves before it has)
            }
        }
        __syncthreads();
        STAMP(23, blockIdx.x == 0 && threadIdx.x == 0);
        if (s_last) skip = 3;   // (gave up: no units, no contacts -- nothing of k_gprep's is trusted; the block still takes its ticket, the step fails)
        __syncthreads();
    }
    for (int i = threadIdx.x; i < US_NCAND; i += 256) s_acc[i] = 0;
    {
        const int* src = reinterpret_cast<const int*>(s2.uset);
        int* dst = reinterpret_cast<int*>(&s_U);
        for (int i = threadIdx.x; i < (int)(sizeof(USet) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const USet& U = s_U;
    const int TL = U.tile;                                        // fragments per tile: 64, or 32 (several sub-fragments per bin)
    const bool halves = TL == 32;                                 // lanes 0-31 and 32-63 hold the SAME 32 fragments and take two fragments of the segment at a time
    const int xi = lane & (TL - 1), half = halves ? (lane >> 5) : 0;
    const int seg_cap = MULTI ? (halves ? 4 : 2) : SEG;           // fragments of a unit's segment
    const unsigned long long nq_total = counters[2];              // written by k_scan, an earlier kernel on the stream
    const unsigned long long n_units = min(*list_n, sa.list_cap); // written by k_gprep (ordered by an event)
    const float nfpb = sa.nfpb;
    const Par par = sa.par;
    const bool quirk = sa.quirk != 0;
    const int reach_bp = sa.reach_bp;
    const float norm_u = fa.norm_u;
    STile2* const tile = s_tile[wib];
    float (*const cy)[4] = s_cy[wib];
    float (*const exo)[64] = s_exo[wib];
    STAMP(17, blockIdx.x == 0 && threadIdx.x == 0);
    STAMP_FBLK(0, threadIdx.x == 0);   // (debug build, per block: past the prologue | wave 0's first unit: entries read | fragments and classes in | at the class loop)
    // ---- (1) the listed units.  The list holds entries of at most seg_unit fragments of the segment side; a wave takes m neighbouring
    // entries -- merged into one unit where they continue each other (many entries: fewer, longer units, the lanes' side loaded once) -- or
    // one entry is dealt to R waves that share its classes (few entries: a unit's depth, all classes of a pair one after the other, is what a
    // step of a few hundred fragments waits for)
    int mrg = 1, rep_n = 1;
    {
        const int m_max = seg_cap / s2.seg_unit;
        while (mrg < m_max && n_units / (unsigned long long)(2 * mrg) >= s2.target) mrg <<= 1;
        // (sharing a unit costs every sharing wave the current layout's values again: only while there are fewer units than waves)
        // ... and never more than one unit per wave through sharing (a second unit is a second chain of dependent loads and passes)
        if (mrg == 1) while (2 * rep_n <= s2.rep_max && n_units * (unsigned long long)(2 * rep_n) <= (unsigned long long)n_waves) rep_n <<= 1;
    }
    S2_COUNT(7, (blockIdx.x == 0 && threadIdx.x == 0) ? (unsigned long long)n_units * 1000000ull + (unsigned long long)(mrg * 100 + rep_n) : 0ull);
    // A step with fewer units than waves (contigs of a few hundred bins): the waves are dealt to the units BY THEIR CLASSES.  A unit's cost
    // is its piece pair's classes -- 0 to 60 of them, one pass over the unit each -- and with the same number of waves for every unit the
    // step waited for the waves of the heaviest one (C2 stand-in, 7 contigs: 348 units, wave 0 of the blocks done with its units after 17 us
    // on average, the last one after 47).  Every block derives the same plan: classes per unit from the entries' pair index, q classes per
    // wave so that the plan fits the grid, units u's waves = [s_pref[u], s_pref[u + 1]).
    const bool bal = !(skip & 1) && mrg == 1 && n_units > 0ull && n_units <= (unsigned long long)S2_BAL_MAX && n_units * 2ull <= (unsigned long long)n_waves && s2.rep_max > 1;
    if (bal) {
        const int nu = (int)n_units, t = threadIdx.x;
        int ncu[4], mine = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int u = 4 * t + i;
            ncu[i] = 0;
            if (u < nu) { const int nc_u = s2.cls_n[uunit_pair(list[u])]; ncu[i] = nc_u > 1 ? nc_u : 1; }
            mine += ncu[i];
        }
        s_bsum[t] = mine;
        __syncthreads();
        if (t < 64) wave_excl_scan(s_bsum, s_boff, 256);
        __syncthreads();
        // (an eighth of the grid stays free for the queued contacts: the waves without a unit price them while the others are at their units; with
        // every wave at a unit the contacts waited for the units, 10 us more on the C2 stand-in)
        const int c_res = n_waves / 8;
        const int total = s_boff[256], spare = max(n_waves - c_res - nu, 1);
        const int q = spare > 0 ? (total + spare - 1) / spare : total;     // sum of ceil(nc / q) <= total / q + units <= waves
        __syncthreads();
        mine = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { ncu[i] = ncu[i] > 0 ? (ncu[i] + q - 1) / (q > 0 ? q : 1) : 0; mine += ncu[i]; }
        s_bsum[t] = mine;
        __syncthreads();
        if (t < 64) wave_excl_scan(s_bsum, s_boff, 256);
        __syncthreads();
        int at = s_boff[t];
#pragma unroll
        for (int i = 0; i < 4; i++) { const int u = 4 * t + i; if (u < nu) s_pref[u] = at; at += ncu[i]; }
        if (t == 0) s_pref[nu] = s_boff[256];
        __syncthreads();
    }
    const unsigned long long n_virtual = (skip & 1) ? 0ull : (bal ? (unsigned long long)s_pref[(int)n_units] :
                                         ((n_units + (unsigned long long)mrg - 1ull) / (unsigned long long)mrg) * (unsigned long long)rep_n);
    // a wave's first unit is its own number; the others it draws from a counter -- the draw goes out when a unit is started and is read
    // when it is finished (units differ in cost by the classes of their piece pair and by the window: dealt round robin, the waves that
    // got the heavy ones finished a fifth of the kernel after the others)
    const bool draw = n_virtual >= (unsigned long long)s2.draw_min * (unsigned long long)n_waves;   // (few units per wave: dealt round robin -- a draw is a round trip)
    const int rep_sh = __ffs(rep_n) - 1;
    for (unsigned long long v = (unsigned long long)wave; v < n_virtual;) {
      unsigned long long drawn = 0;
      if (draw && lane == 0) drawn = atomicAdd(s2.next, 1ull);
      int rep_r = (int)(v & (unsigned long long)(rep_n - 1));                             // (rep_n, mrg: powers of two)
      unsigned long long e0 = (v >> rep_sh) * (unsigned long long)mrg;
      if (bal) {   // the unit whose waves hold number v
          int lo = 0, hi = (int)n_units - 1;
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= (int)v) lo = mid; else hi = mid - 1; }
          e0 = (unsigned long long)lo;
          rep_r = (int)v - s_pref[lo];
          rep_n = s_pref[lo + 1] - s_pref[lo];
      }
      const int m_in = (int)(min(n_units, e0 + (unsigned long long)mrg) - e0);
      // the group's entries, one per lane (one round trip); runs of entries that continue each other become one unit
      if (lane < m_in) s_ent[wib][lane] = list[e0 + (unsigned long long)lane];   // (kept in LDS: live across a unit they cost registers the pair loop needs)
      WAVE_LDS_SYNC();
      STAMP_FBLK(1, threadIdx.x == 0 && v == (unsigned long long)wave && s_ent[wib][0] != ~0ull);   // (the condition reads the entry: the stamp waits for it)
      int pos = 0;
      while (pos < m_in) {
        const unsigned long long my = lane < m_in ? s_ent[wib][lane] : 0ull;
        const int my_key = (int)(my & 0xffffffffull), my_j0 = (int)((my >> 32) & 63ull), my_cnt = (int)((my >> 38) & 63ull), my_lf = (int)((my >> 44) & 1ull);
        const int key0 = __shfl(my_key, pos, 64), j0 = rfl(__shfl(my_j0, pos, 64)), lf = rfl(__shfl(my_lf, pos, 64));
        const bool cont = lane >= pos && lane < m_in && my_key == key0 && my_lf == lf && my_j0 == j0 + (lane - pos) * s2.seg_unit;
        const unsigned long long bl = __ballot(cont) >> pos;          // bit i: entry pos + i continues the run (bit 0: the entry itself)
        int run = (int)__ffsll((long long)~bl) - 1;
        run = min(run, max(seg_cap / s2.seg_unit, 1));
        const int cnt = rfl((run - 1) * s2.seg_unit + __shfl(my_cnt, pos + run - 1, 64));
        const int ti = rfl(key0 & 0xffff), tj = rfl((key0 >> 16) & 0xffff);
        pos += run;
        int offi, offj;
        const int g = rfl(utile_piece(U, ti, offi)), h = rfl(utile_piece(U, tj, offj));
        offi = rfl(offi); offj = rfl(offj);
        const int gl = lf ? g : h, gs = lf ? h : g, offl = lf ? offi : offj, offs = lf ? offj : offi;
        const UPiece& PL = U.p[gl];
        const UPiece& PS = U.p[gs];
        const UContig& CL = U.c[PL.contig];
        const UContig& CS = U.c[PS.contig];
        // (the pair's first CLS_CHUNK class records and their number: requested NOW, with the fragments' records -- they depend on the piece pair
        // alone.  Records behind the pair's last one are read for nothing: the table holds US_NCAND per pair)
        const int pair = upair_index(g, h);
        const GClass* __restrict__ cg = s2.cls + (size_t)pair * US_NCAND;
        const uint4 cls_pre0 = reinterpret_cast<const uint4*>(cg)[lane], cls_pre1 = reinterpret_cast<const uint4*>(cg)[64 + lane];
        const int nc_pre = s2.cls_n[pair];
        const int nl = rfl(min(TL, PL.n - offl * TL));
        const bool has_l = xi < nl;
        const int fl = has_l ? sa.perm[CL.base + PL.lo + offl * TL + xi] : 0;
        Geo gL = {0, 0, 0, 0};
        Stat stL = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
        if (has_l) { gL = geo[fl]; stL = stat[fl]; }
        if (lane < cnt) {
            STile2 y;
            y.frag = sa.perm[CS.base + PS.lo + offs * TL + j0 + lane];
            const Geo gy = geo[y.frag];
            y.start_bp = gy.start_bp; y.len_bp = gy.len_bp; y.flags = gy.flags; y.st = stat[y.frag];
            tile[lane] = y;
        }
        WAVE_LDS_SYNC();
        STAMP_FBLK(2, threadIdx.x == 0 && v == (unsigned long long)wave && pos == run && tile[0].frag >= 0 && gL.start_bp != -12345);   // (reads what was loaded, likewise)
        const bool cis_old = rfl(PL.contig) == rfl(PS.contig);
        const int circ_old = rfl(CL.circ), lbp_old = rfl(CL.lbp);
        // old bp extents of the two sides (positions grow with start_bp inside a contig)
        const int xs = rfl(gL.start_bp), xe = __builtin_amdgcn_readlane(gL.start_bp + gL.len_bp, nl - 1);
        const int ys = rfl(tile[0].start_bp), ye = rfl(tile[cnt - 1].start_bp + tile[cnt - 1].len_bp);
        const bool near_old = cis_old && max(ys - xe, xs - ye) <= reach_bp;
        bool always = false;
        if (quirk) always = __ballot((has_l && !stat_uniform(stL)) || (lane < cnt && !stat_uniform(tile[lane < cnt ? lane : 0].st))) != 0ull;
        const int nc = rfl(nc_pre);
        // the pair's class records: into LDS in one round trip (read one by one where they are needed, each was a round trip of its own -- with
        // a dozen classes per unit, most of a small step)
        GClass* const cp = s_cls[wib];
        int c_base = 0;                      // cp[i] = class c_base + i, CLS_CHUNK of them
        auto stage = [&](int c0) {
            WAVE_LDS_SYNC();
            const uint4* src = reinterpret_cast<const uint4*>(cg + c0);
            uint4* dst = reinterpret_cast<uint4*>(cp);
            const int n4 = min(nc - c0, CLS_CHUNK) * 4;
            for (int i = lane; i < n4; i += 64) dst[i] = src[i];
            c_base = c0;
            WAVE_LDS_SYNC();
        };
        {   // (the first chunk was requested with the unit's fragments, above: one round trip for both)
            uint4* dst = reinterpret_cast<uint4*>(cp);
            dst[lane] = cls_pre0; dst[64 + lane] = cls_pre1;
            WAVE_LDS_SYNC();
        }
        // is there anything to price?  (a listed unit has, but for the finer extents of its segment)
        bool any = false;
        for (int c = rep_r; c < nc && !any; c += rep_n) {
            if (c < c_base || c >= c_base + CLS_CHUNK) stage(c);
            const GClass& cr = cp[c - c_base];
            if (!(cr.w0 | cr.w1 | (unsigned long long)cr.w2)) continue;   // (a class of neighbours priced by the table kernel)
            if (near_old || always) { any = true; break; }
            const unsigned flags = (unsigned)rfl((int)cr.flags);
            if (!(flags & 4u)) continue;
            const int offx = rfl(cr.offx), offy = rfl(cr.offy);
            const int sig_l = lf ? (flags & 1u) : ((flags >> 1) & 1u), sig_s = lf ? ((flags >> 1) & 1u) : (flags & 1u);
            const int off_l = lf ? offx : offy, off_s = lf ? offy : offx;
            const int xs2 = sig_l ? xs + off_l : off_l - xe, xe2 = sig_l ? xe + off_l : off_l - xs;
            const int ys2 = sig_s ? ys + off_s : off_s - ye, ye2 = sig_s ? ye + off_s : off_s - ys;
            if (max(ys2 - xe2, xs2 - ye2) <= reach_bp) any = true;
        }
        if (!any) { WAVE_LDS_SYNC(); continue; }
        const bool fwdL = (gL.flags & 1) != 0;
        const bool diag = ti == tj;
        unsigned vmask = 0;   // bit j: this lane's fragment and fragment j of the segment are a pair to price (every unordered pair once; never a
                              // bin with itself; copies of repeated bins -- no sub-fragments here -- are priced by k_rep_delta)
        for (int j = 0; j < cnt; j++)
            if (has_l && stL.n > 0 && tile[j].st.n > 0 && !(diag && !(xi < j0 + j))) vmask |= 1u << j;
        STAMP_FBLK(3, threadIdx.x == 0 && v == (unsigned long long)wave && pos == run);
        for (int c = -1; c < nc; c = c < 0 ? rep_r : c + rep_n) {     // c = -1: the current layout (its values are kept in LDS), then this wave's classes
            bool cis = cis_old;
            int circ = cis_old ? circ_old : 0, lbp = lbp_old, sig_l = 1, sig_s = 1, off_l = 0, off_s = 0;
            unsigned long long m0 = 0, m1 = 0;
            unsigned m2 = 0;
            if (c >= 0) {
                if (c < c_base || c >= c_base + CLS_CHUNK) stage(c);
                const GClass& cr = cp[c - c_base];
                const unsigned flags = (unsigned)rfl((int)cr.flags);
                const int offx = rfl(cr.offx), offy = rfl(cr.offy);
                cis = (flags & 4u) != 0; circ = (flags >> 3) & 1u; lbp = rfl(cr.lbp);
                sig_l = lf ? (flags & 1u) : ((flags >> 1) & 1u); sig_s = lf ? ((flags >> 1) & 1u) : (flags & 1u);
                off_l = lf ? offx : offy; off_s = lf ? offy : offx;
                if (!(near_old || always)) {   // the trans value both times, slot by slot: exactly zero
                    if (!cis) continue;
                    const int xs2 = sig_l ? xs + off_l : off_l - xe, xe2 = sig_l ? xe + off_l : off_l - xs;
                    const int ys2 = sig_s ? ys + off_s : off_s - ye, ye2 = sig_s ? ye + off_s : off_s - ys;
                    if (max(ys2 - xe2, xs2 - ye2) > reach_bp) continue;
                }
                const unsigned long long mm0 = cr.w0, mm1 = cr.w1;
                m0 = ((unsigned long long)(unsigned)rfl((int)(mm0 >> 32)) << 32) | (unsigned long long)(unsigned)rfl((int)mm0);
                m1 = ((unsigned long long)(unsigned)rfl((int)(mm1 >> 32)) << 32) | (unsigned long long)(unsigned)rfl((int)mm1);
                m2 = (unsigned)rfl((int)cr.w2);
                if (!(m0 | m1 | (unsigned long long)m2)) continue;
            }
            const float s_tot = (float)lbp / 1000.0f;
            // the lanes' side in this layout
            const int startL = gclass_start(sig_l, off_l, gL.start_bp, gL.len_bp);
            const bool fwdLn = fwdL == (sig_l != 0);
            float cl[NS];
#pragma unroll
            for (int a = 0; a < NS; a++) cl[a] = a < stL.n ? centre_kb(startL, fwdLn, stL, a) : 0.0f;
            // the segment side: one lane per fragment
            if (lane < cnt) {
                const STile2& y = tile[lane];
                const int startS = gclass_start(sig_s, off_s, y.start_bp, y.len_bp);
                const bool fwdSn = ((y.flags & 1) != 0) == (sig_s != 0);
#pragma unroll
                for (int b = 0; b < NS; b++) cy[lane][b] = b < y.st.n ? centre_kb(startS, fwdSn, y.st, b) : 0.0f;
                cy[lane][3] = fwdSn ? 1.0f : 0.0f;
            }
            WAVE_LDS_SYNC();
            S2_COUNT(5, lane == 0 ? 1 : 0);                                   // (unit, layout) passes
            S2_COUNT(6, __popc(vmask));                                      // fragment pairs priced in them
            long long accq = 0;
            auto add_pair = [&](int j, double acc) {   // one fragment pair of the class: rounded to Q once
                if ((vmask >> j) & 1u) {
                    const long long q1 = to_q_fast(acc);
                    if (q1 == Q_BAD) s2_coarse_mask(fa.acc, counters + NF_OFF, m0, m1, m2, acc); else accq += q1;
                }
            };
            // the same without branches for the common case (to_q_fast: |v| < 2^30 is v = hi + lo, hi = rint(v), both parts int32; summed apart,
            // joined behind the loop); a larger term takes the slow way
            long long acc_hi = 0, acc_lo = 0;
            auto add_pair_fast = [&](int j, double v, bool mine = true) {
                const bool ok = mine && ((vmask >> j) & 1u) != 0;
                const bool big = ok && !(fabs(v) < 1073741824.0);
                const double hi = rint(v);
                const int ih = __double2int_rn(hi), il = __double2int_rn((v - hi) * Q_SCALE);
                const bool take = ok && !big;
                acc_hi += take ? (long long)ih : 0ll;
                acc_lo += take ? (long long)il : 0ll;
                if (__builtin_expect(big, 0)) {
                    const long long q1 = to_q(v);
                    if (q1 == Q_BAD) s2_coarse_mask(fa.acc, counters + NF_OFF, m0, m1, m2, v); else accq += q1;
                }
            };
            if (!MULTI && norm_u >= 0.0f && !(quirk && !cis)) {
                // one sub-fragment per bin, one RF count: nothing per pair but the model
                // (tiles of 32: this half of the wave takes fragment jb + half of the segment; its values of the current layout in row jb / 2)
                const int jstep = halves ? 2 : 1;
                if (!cis) {
                    const float ex = par.v_inter * norm_u;
                    for (int jb = 0; jb < cnt; jb += jstep) {
                        const int j_mine = jb + half, j = j_mine < cnt ? j_mine : cnt - 1, jrow = halves ? (jb >> 1) : jb;
                        if (c < 0) exo[jrow][lane] = ex;
                        else add_pair_fast(j, (double)exo[jrow][lane] - (double)ex, j_mine < cnt);
                    }
                } else {
                    for (int jb = 0; jb < cnt; jb += jstep) {
                        const int j_mine = jb + half, j = j_mine < cnt ? j_mine : cnt - 1, jrow = halves ? (jb >> 1) : jb;
                        const float c_s = cy[j][0];
                        const float ex = s2_ex(true, lf ? cl[0] : c_s, lf ? c_s : cl[0], norm_u, circ, s_tot, par);
                        if (c < 0) exo[jrow][lane] = ex;
                        else add_pair_fast(j, (double)exo[jrow][lane] - (double)ex, j_mine < cnt);
#if defined(GRAAL_STAMPS) && defined(GRAAL_S2_COUNTS)
                        {   // (round-4 review item 6: how many (pair, class) evaluations hand the model the float32 inputs of the CURRENT layout?)
                            const float s_dbg = fabsf((lf ? c_s : cl[0]) - (lf ? cl[0] : c_s));
                            if (c < 0) s_sold[wib][j][lane] = s_dbg;
                            else {
                                const bool valid = ((vmask >> j) & 1u) != 0;
                                const bool same = valid && cis_old && circ == circ_old && (!circ || lbp == lbp_old) && s_dbg == s_sold[wib][j][lane];
                                const int n_valid = __popcll(__ballot(valid)), n_same = __popcll(__ballot(same));
                                if (lane == 0 && n_valid > 0) {
                                    atomicAdd(&g_s2eq[0], (unsigned long long)n_valid);
                                    atomicAdd(&g_s2eq[1], (unsigned long long)n_same);
                                    atomicAdd(&g_s2eq[2], n_same == n_valid ? (unsigned long long)n_valid : 0ull);      // evaluations in passes a whole wave could skip
                                    atomicAdd(&g_s2eq[3], 2 * n_same >= n_valid ? (unsigned long long)n_same : 0ull);   // ... that lane compaction could drop (>= half agree)
                                    atomicAdd(&g_s2eq[4], 1ull);                                                          // (wave, segment fragment, class) passes
                                    atomicAdd(&g_s2eq[5], !cis_old ? (unsigned long long)n_valid : 0ull);                 // pairs that are trans in the current layout
                                }
                            }
                        }
#endif
                    }
                }
                accq += acc_hi * (1ll << 30) + acc_lo;
            } else
            for (int jb = 0; jb < cnt; jb += (halves ? 2 : 1)) {
                // (tiles of 32: this half of the wave takes fragment jb + half of the segment; a lane without one computes on the last and adds nothing)
                const int j_mine = jb + half, j = j_mine < cnt ? j_mine : cnt - 1;
                const bool j_ok = j_mine < cnt;
                const int jrow = halves ? (jb >> 1) : jb;                  // the lane's row of current-layout values
                const STile2& y = tile[j];
                const int ny = y.st.n, fs = y.frag;
                const bool fwdSn = cy[j][3] != 0.0f;
                double acc = 0.0;
                // slot pairs in the order of the O(m^2) kernel: the FIRST fragment's slots outside (first = earlier in the union's order:
                // tile ti's side), the second's inside
                const int n_first = lf ? stL.n : ny, n_second = lf ? ny : stL.n;
                for (int a = 0; a < NS; a++)
                    for (int b = 0; b < NS; b++) {
                        if (MULTI && !(a < n_first && b < n_second)) continue;
                        const int sl = lf ? a : b, ss = lf ? b : a;       // the lane fragment's slot, the segment fragment's slot
                        const float c_l = NS == 1 ? cl[0] : sel3(cl[0], cl[NS > 1 ? 1 : 0], cl[NS > 2 ? 2 : 0], sl);
                        const float c_s = cy[j][ss];
                        int ax = stat_accu(stL, sl), ay = stat_accu(y.st, ss);
                        if (quirk && !cis) {   // the reference's trans-branch indexing: a reversed lower-id bin is priced with its LAST RF count
                            if (fl < fs) { if (!fwdLn) ax = stat_accu(stL, stL.n - 1); }
                            else if (!fwdSn) ay = stat_accu(y.st, ny - 1);
                        }
                        const float norm = norm_u >= 0.0f ? norm_u : (float)(ax * ay) / nfpb;
                        const float ex = s2_ex(cis, lf ? c_l : c_s, lf ? c_s : c_l, norm, circ, s_tot, par);
                        float* const slot = &exo[jrow * NSP + a * NS + b][lane];
                        if (c < 0) *slot = ex;
                        else acc += (double)*slot - (double)ex;
                    }
                if (c >= 0 && j_ok) add_pair(j, acc);
            }
            if (c >= 0) {
                const long long qs = wave_sum_ll(accq);
                const long long v = ((long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(qs >> 32)) << 32) | (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)qs);
                if (v != 0) s2_add_mask(s_acc, m0, m1, m2, v, lane);
            }
            WAVE_LDS_SYNC();   // (the next pass writes the segment's centres again)
        }
      }
      WAVE_LDS_SYNC();
      if (draw) v = (unsigned long long)n_waves + (((unsigned long long)(unsigned)rfl((int)(drawn >> 32)) << 32) | (unsigned long long)(unsigned)rfl((int)drawn));
      else v += (unsigned long long)n_waves;
    }
    STAMP_MAX(18, lane == 0);
    // ---- (2) the queued contacts (the scan queued every contact with both ends in some neighbour's set): lane = contact, one
    // evaluation per class of its piece pair
    {
        QSrc qs;
        qs.queue = fa.queue; qs.geo2 = reinterpret_cast<const int2*>(geo); qs.cnt = fa.cnt; qs.keys = nullptr; qs.live = 0; qs.K = K;
        qs.seq = (unsigned)seq; qs.concurrent = 0; qs.multi = fa.multi;
        // (a step with fewer units than waves: the waves WITHOUT a unit take the contacts -- if their lanes can hold them all -- and price them while
        // the others are still at their units; else every wave takes its share behind its units, from the other end of the grid)
        const unsigned long long n_idle = n_virtual < (unsigned long long)n_waves ? (unsigned long long)n_waves - n_virtual : 0ull;
        const bool idle_take = n_idle * 64ull >= nq_total && nq_total > 0ull;
        const unsigned long long c_waves = idle_take ? n_idle : (unsigned long long)n_waves;
        // lanes per contact: a contact's classes -- a dozen or two, each a record fetched from memory and two evaluations -- are a chain of that
        // many round trips in ONE lane; with few contacts per wave 4 or 16 lanes share a contact's classes (lane s takes the classes s, s + LPC, ...)
        const int lpc = nq_total > 16ull * c_waves ? 1 : (nq_total > 4ull * c_waves ? 4 : 16);
        const int per_wave = 64 / lpc;
        const unsigned long long c_first = idle_take ? (unsigned long long)wave - n_virtual : (unsigned long long)(n_waves - 1 - wave);
        const unsigned long long c_end = ((skip & 2) || (idle_take && (unsigned long long)wave < n_virtual)) ? 0ull : nq_total;
        for (unsigned long long b0 = c_first * (unsigned long long)per_wave; b0 < c_end; b0 += c_waves * (unsigned long long)per_wave) {
            const unsigned long long e = b0 + (unsigned long long)(lane / lpc);
            const int sub = lane % lpc;
            if (e >= nq_total) continue;
            const QEntry qe = q_fetch(qs, e, counters + 6);
            if (qe.fx < 0) continue;
            const int fx = qe.fx, fy = qe.fy, slx = qe.slots & 3, sly = (qe.slots >> 2) & 3;
            const Geo gx = geo[fx], gy = geo[fy];
            const Stat sx = stat[fx], sy = stat[fy];
            const int cnt_bits = fa.cnt[qe.idx];
            const int pgx = ufrag_piece(U, gx.id_c, geo_pos(gx.flags)), pgy = ufrag_piece(U, gy.id_c, geo_pos(gy.flags));
            if (pgx < 0 || pgy < 0) continue;
            const bool x_low = pgx <= pgy;                 // the class records name the lower piece "x"
            const int pair = x_low ? upair_index(pgx, pgy) : upair_index(pgy, pgx);
            const int nc = s2.cls_n[pair];
            if (sub >= nc) continue;
            const GClass* __restrict__ cp = s2.cls + (size_t)pair * US_NCAND;
            GClass nx = cp[sub];                            // (the next class's record is on its way while this one is priced)
            const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? sa.lcontbp[fx] : 0), Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? sa.lcontbp[fy] : 0);
            const float ex_old = ex_pair_ref(X0, sx, slx, fx, Y0, sy, sly, fy, nfpb, par, quirk);
            const double ln_old = mm_ln(ex_old), ob = (double)__int_as_float(cnt_bits);
            for (int c = sub; c < nc; c += lpc) {
                const GClass cr = nx;
                if (c + lpc < nc) nx = cp[c + lpc];
                const int sig_x = x_low ? (cr.flags & 1u) : ((cr.flags >> 1) & 1u), sig_y = x_low ? ((cr.flags >> 1) & 1u) : (cr.flags & 1u);
                const int off_x = x_low ? cr.offx : cr.offy, off_y = x_low ? cr.offy : cr.offx;
                End X, Y;
                const bool cis = (cr.flags & 4u) != 0;
                X.label = 0; Y.label = cis ? 0 : 1;
                X.start_bp = gclass_start(sig_x, off_x, gx.start_bp, gx.len_bp); Y.start_bp = gclass_start(sig_y, off_y, gy.start_bp, gy.len_bp);
                X.fwd = ((gx.flags & 1) != 0) == (sig_x != 0); Y.fwd = ((gy.flags & 1) != 0) == (sig_y != 0);
                X.circ = Y.circ = (cr.flags >> 3) & 1u; X.lbp = Y.lbp = cr.lbp;
                const float ex_new = ex_pair_ref(X, sx, slx, fx, Y, sy, sly, fy, nfpb, par, quirk);
                if (ex_new == ex_old) continue;
                const long long qv = to_q(ob * (mm_ln(ex_new) - ln_old));
                unsigned long long mm[3] = {cr.m0, cr.m1, (unsigned long long)cr.m2};
                for (int r = 0; r < 3; r++) {
                    unsigned long long m = mm[r];
                    while (m) {
                        const int cand = 64 * r + __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        if (qv == Q_BAD) nf_flag(counters + NF_OFF, cand / N_OPS, cand % N_OPS);
                        else if (qv != 0) atomicAdd((unsigned long long*)&s_acc[cand], (unsigned long long)qv);
                    }
                }
            }
        }
    }
    STAMP_MAX(19, lane == 0);
    __syncthreads();
    for (int i = threadIdx.x; i < K * N_OPS; i += 256) {
        const long long v = s_acc[i];
        if (v != 0) atomicAdd((unsigned long long*)&fa.acc[i], (unsigned long long)v);
    }
    if (threadIdx.x == 255 && blockIdx.x == 0) atomicAdd(&counters[1], n_units);
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(&counters[5], 1ull);
        s_last = (ticket == (unsigned long long)gridDim.x - 1ull);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) { *list_n = 0; *s2.next = 0; }   // (every block has read them: the list is empty again for the next step)
    hand_out(fa.acc, counters, fa.sync, K, d_q_out, host_res, seq);
    STAMP(20, threadIdx.x == 0);
}
