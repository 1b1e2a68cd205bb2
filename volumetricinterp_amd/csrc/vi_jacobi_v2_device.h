// K3, role-separated (round 4): the same Jacobi iteration as jacobi_system (vi_jacobi_device.h) - same ordering, same rotation
// formulas, same arithmetic per element, THE SAME BITS (tools/ab_jacobi_bits.py, tests/test_gpu_search_stages.py) - with the work
// of a round split between a SET-UP WAVE and UPDATE WAVES.
//
// What a round of jacobi_system costs (4300 cycles at N = 144; tools/microbench/lds_exchange.hip, tools/exp_stamps.py): 1600 for
// the rotation set-up in wave 0 while nine waves wait, ~1300 of block arithmetic, and an exchange through LDS whose store burst
// alone - 160 ds_write_b64 on a store path of ~85 B/clk - is 1100, all of it serial: set-up | barrier | update + stores | barrier.
// The set-up needs only the diagonal blocks.  Here
//   * wave 0 holds NO off-diagonal block: it keeps the diagonal blocks and the right-hand side of all matches in registers for
//     the whole run, computes the rotations, and assembles the next round's diagonal blocks from its own lanes (the two diagonal
//     2 x 2 parts and the halves of y travel by single-lane DPP shifts - done while the update waves are still working) and from
//     a four-double MAILBOX per match in LDS (the cross part);
//   * the update waves hold one super-block per thread (blockDim = 64 + the threads jacobi_system would use); the M blocks that
//     hold a cross part belong to the FIRST update wave (thread 64 + a' holds the block of match a' - the block -> thread map is
//     a permutation built at the start of a solve), which runs its update at raised priority and posts the mailbox while the other
//     waves are still computing: wave 0 reads it from an LDS pipe that is still quiet and starts the next set-up under the other
//     waves' arithmetic and store burst;
//   * two monotonic counters in LDS (mailbox posted; stores of the round landed - one count per wave; bounded polling) and ONE
//     workgroup barrier per round (rotations published, blocks fetched) take the place of the two barriers;
//   * the two roles run the same loop skeleton as two separate loops, so that each gets a register allocation of its own: 155
//     VGPRs, nothing spilled (one loop with role branches spilled the sixteen store addresses: 3500 cycles per round).
// Measured (MI355X, 64 cold systems of 25 sweeps): N = 144 4.22 -> 3.70 ms (-12.5 %), -4 ... -12 % over N = 93 ... 149, break-even at
// N = 72, +2 % at N = 32: used from 24 matches on (vi_jacobi_use_v2).  Stamps (tools/exp_stamps_v2.py): set-up wave 1830 + copies 240
// + hand-over 590 per round, update waves fetch 660 + update 2200 + stores 280 - the update waves the longer path at ~3360 stamped
// cycles where the two-barrier kernel has 4300 serial ones.
// What did NOT work on the way is under tools/experiments/ (reading the diagonal planes back from LDS; one loop for both roles;
// holding the stores back until the mailbox is read).
#pragma once
#include "vi_jacobi_device.h"

namespace {

// LDS operations of a wave execute in order, so "my stores, then the count" needs only the LDS counter to drain before the
// add - not the C++ release ordering, which also waits for the global stores in flight (the rotation log, a microsecond).
__device__ __forceinline__ void lds_signal(int* w, int n)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(w, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Wave-uniform bounded wait (~2^22 polls: seconds): a counter that never arrives ends the wait instead of hanging the GPU;
// the solve then reports "not converged" through `bad`.
template <bool SLEEP = true>
__device__ __forceinline__ void lds_wait(int* w, int target, int& bad)
{
    const int tgt = __builtin_amdgcn_readfirstlane(target);
    int spins = 0;
#pragma nounroll
    for (;;) {
        const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= tgt) break;
        if (++spins > (1 << 22)) { bad = 1; break; }
        if (SLEEP) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// blockDim.x = 64 + NU, NU = the number of super-blocks rounded up to whole waves; requires 3 <= M <= 64 (N >= 9).
// LDS: vi_jacobi_v2_lds_bytes(N).
__device__ __forceinline__ void jacobi_system_v2(
    unsigned char* lds_raw, int N, const double* __restrict__ Xs, double sc, const double* __restrict__ ys, double rcond,
    double abs_floor, double* __restrict__ Cs, int* __restrict__ rank_s, double2* __restrict__ logp, int max_sweeps,
    int* __restrict__ sweeps_s, double* __restrict__ lam_s, int lam_raw, int* __restrict__ nround_s,
    unsigned long long* __restrict__ round_acc, double conv_tol = 0.0)
{
    const int NT = blockDim.x;
    const int Np = (N + 3) & ~3;
    const int m = Np >> 1;
    const int M = Np >> 2;
    const int nsb = (M * (M - 1)) >> 1;
    const int ntri = 16 * nsb + 10 * M;
    const int dg = 16 * nsb;
    double* A = reinterpret_cast<double*>(lds_raw);                          // [ntri] the slot-indexed image, as in jacobi_system
    double* yv = A + ntri;                                                   // [2][Np]
    double2* cs0 = reinterpret_cast<double2*>(yv + 2 * Np);                  // [2][4][M] rotations of a round, double-buffered
    double* nd = reinterpret_cast<double*>(cs0 + 8 * M);                     // [16]
    double* mb = nd + 16;                                                    // [4][M] mailbox: cross parts of the next diagonal blocks
    int* sync = reinterpret_cast<int*>(mb + 4 * M);                          // [0] mailbox posted [2] stores of the round landed
    int* dtab = sync + 4;                                                    // [14][M] permuted destinations of diagonal blocks and y

    const int tid = threadIdx.x;
    const int nw = NT >> 6;
    const bool setup = tid < 64;                                             // wave 0

    // ---- update threads: my super-block (a < b), k = b(b-1)/2 + a = tid - 64
    int ka = 0, kb = 1, ksrc = 0, dst[16];
    bool live = false;
    int serves = -1, q0 = 0;            // my block holds the cross part of the next diagonal block of match `serves`, at q0 + {0,1,4,5}
    // ---- which update thread owns which super-block: the M blocks that hold the cross part of a next diagonal block go to the
    //      first M update threads (wave 1, which runs its update at raised priority and posts the mailbox early), thread 64 + a'
    //      holding the block of match a'; the others follow in natural order.  The table lives where the image goes afterwards.
    {
        int* otab = reinterpret_cast<int*>(A);                       // [NT] block of update thread u
        int* wcnt = otab + NT;                                       // [nw] non-designated blocks per wave
        const int lane = tid & 63, wv = tid >> 6;
        const int kn = tid - 64;                                     // natural block of this thread
        bool isd = false;
        int sv = 0;
        if (!setup && kn < nsb) {
            int bb = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)kn)) * 0.5f);
            while ((bb * (bb - 1)) / 2 > kn) --bb;
            while (((bb + 1) * bb) / 2 <= kn) ++bb;
            const int aa = kn - (bb * (bb - 1)) / 2;
            if (aa == 0 && bb == 1) { isd = true; sv = 0; }
            else if (aa == 0 && bb == 2) { isd = true; sv = 1; }
            else if (aa == M - 2 && bb == M - 1) { isd = true; sv = M - 1; }
            else if (aa >= 1 && bb == aa + 2) { isd = true; sv = aa + 1; }
        }
        const bool plain = !setup && kn < nsb && !isd;
        const unsigned long long mn = __ballot(plain);
        if (lane == 0) wcnt[wv] = __popcll(mn);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wv; ++w) before += wcnt[w];
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (isd) otab[sv] = kn;
        if (plain) otab[M + before + __popcll(mn & below)] = kn;
        __syncthreads();
        if (!setup && kn < nsb) ksrc = otab[kn];
        __syncthreads();                                             // the table is read: the image may be loaded over it
    }
    if (!setup) {
        const int ko = tid - 64;
        live = ko < nsb;
        const int k = live ? ksrc : 0;
        int b = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
        while ((b * (b - 1)) / 2 > k) --b;
        while (((b + 1) * b) / 2 <= k) ++b;
        const int a = k - (b * (b - 1)) / 2;
        ka = live ? a : 0;
        kb = live ? b : 1;
        ksrc = k;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[4 * r + c] = tri4(slot_next(4 * ka + r, M), slot_next(4 * kb + c, M), M);
        // Units move along the ring (ring_next): the new unit U of match a' comes from (a'-1, U) - from (0, V) for a' = 1, stays
        // for a' = 0 -, the new unit V from (a'+1, V) - from (M-1, U) for a' = M-1.  The cross part of the new diagonal block a'
        // is therefore a 2 x 2 quadrant of ONE old off-diagonal block:
        //   a' = 0: (0,1) rows U x cols V;  a' = 1: (0,2) rows V x cols V;  2 <= a' <= M-2: (a'-1, a'+1) rows U x cols V;
        //   a' = M-1: (M-2, M-1) rows U x cols U.
        if (live) {
            if (a == 0 && b == 1) { serves = 0; q0 = 2; }
            else if (a == 0 && b == 2) { serves = 1; q0 = 10; }
            else if (a == M - 2 && b == M - 1) { serves = M - 1; q0 = 0; }
            else if (a >= 1 && b == a + 2) { serves = a + 1; q0 = 2; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[e] = 0;
        if (tid < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dtab[(10 + r) * M + tid] = slot_next(4 * tid + r, M);
#pragma unroll
                for (int c = r; c < 4; ++c) dtab[j10(r, c) * M + tid] = tri4(slot_next(4 * tid + r, M), slot_next(4 * tid + c, M), M);
            }
        }
    }

    // ---- load (slot s holds original index s) ---------------------------------------------------------
    double mxd = 0.0;
    for (int e = tid; e < Np * Np; e += NT) {
        const int i = e / Np, j = e - i * Np;
        if (j > i) continue;
        const double v = (i < N && j < N) ? Xs[(int64_t)i * N + j] : 0.0;
        A[tri4(i, j, M)] = v;
        if (i == j) mxd = fmax(mxd, fabs(v));
    }
    for (int s = tid; s < Np; s += NT) yv[s] = s < N ? ys[s] : 0.0;
    for (int o = 32; o > 0; o >>= 1) mxd = fmax(mxd, __shfl_xor(mxd, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mxd;
    if (tid == 0) { sync[0] = 0; sync[1] = 0; sync[2] = 0; }
    __syncthreads();
    mxd = 0.0;
    for (int w = 0; w < nw; ++w) mxd = fmax(mxd, nd[w]);
    const double drop = rcond * mxd;
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    const double conv2 = conv_tol > 0.0 ? conv_tol * conv_tol : VI_CONV_FACTOR * VI_CONV_FACTOR * eps2;
    int sweep = 0, ycur = 0;
    bool converged = false;
    int64_t nround = 0;
    int bad = 0;                                   // a counter did not arrive (never seen; ends as "not converged")
    int tgt_mb = 0, tgt_st = 0;        // what the counters read once the round in question is through
    // The two roles run the SAME loop skeleton - rounds, one barrier per round, the two sweep-end votes - as two separate loops, so
    // that the register allocation of each sees only its own state (in one loop with role branches the allocator spilled the
    // sixteen store addresses of the update waves and reloaded them from scratch one per store: 3500 cycles per round).
#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif
    if (setup) {
        // ================= wave 0: diagonal blocks and right-hand side of all matches, in registers from the first round on
        double d[4][4], yy[4];
        bool in_regs = false;
        // permuted destinations of my diagonal block and of y, out of the table once (read next to their use, each would wait
        // for the LDS round trip of its own: 14 x ~110 cycles on the path the whole workgroup waits for)
        int dd[10], yd[4];
#pragma unroll
        for (int j = 0; j < 10; ++j) dd[j] = tid < M ? dtab[j * M + tid] : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) yd[p] = tid < M ? dtab[(10 + p) * M + tid] : 0;
        for (; sweep < max_sweeps; ++sweep) {
            int rotated = 0;
            for (int r = 0; r < m; ++r, ++nround) {
                const bool intra = r == 0;            // round 0 of a sweep: the pairs inside the units, no permutation
                double2* cs = cs0 + (nround & 1) * (4 * M);
                // ---- the rotations of every match, both inner rounds: inputs in registers, outputs to LDS (cs) and to the log
                if (tid < M) {
                    if (!in_regs) {
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int q = p; q < 4; ++q) d[p][q] = d[q][p] = A[dg + j10(p, q) * M + tid];
#pragma unroll
                        for (int p = 0; p < 4; ++p) yy[p] = yv[ycur * Np + 4 * tid + p];
                    }
                    double2 r0, r1, r2 = make_double2(1.0, 0.0), r3 = make_double2(1.0, 0.0);
                    if (intra) {
                        rot_stage<0, 1, 2, 3>(d, yy, drop, abs_floor, rotated, r0, r1);
                    } else {
                        rot_stage<0, 2, 1, 3>(d, yy, drop, abs_floor, rotated, r0, r1);
                        rot_stage<0, 3, 1, 2>(d, yy, drop, abs_floor, rotated, r2, r3);
                    }
                    cs[tid] = r0;
                    cs[M + tid] = r1;
                    cs[2 * M + tid] = r2;
                    cs[3 * M + tid] = r3;
                    double2* lp = logp + nround * (int64_t)(4 * M) + tid;
                    lp[0] = r0;
                    lp[M] = r1;
                    lp[2 * M] = r2;
                    lp[3 * M] = r3;
                }
                VI_STAMP(0);
                __syncthreads();                      // rotations published, blocks fetched
                VI_STAMP(1);
                // ---- what LDS needs of my diagonal block: four of its elements land in an off-diagonal block that an update thread
                //      fetches next round (hence after the barrier, and counted among the stores of the round); the rest and y
                //      are only read by the sweep-end test and the final solve
                if (tid < M) {
                    if (r == m - 1) {
                        // last round of the sweep: the whole block and y (the sweep-end test and the final solve read them)
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            yv[(ycur ^ 1) * Np + yd[p]] = yy[p];
#pragma unroll
                            for (int q = p; q < 4; ++q) A[dd[j10(p, q)]] = d[p][q];
                        }
                    } else if (!intra) {
                        // any other round: only the four elements between my two units - they become part of an off-diagonal block
                        A[dd[j10(0, 2)]] = d[0][2];
                        A[dd[j10(0, 3)]] = d[0][3];
                        A[dd[j10(1, 2)]] = d[1][2];
                        A[dd[j10(1, 3)]] = d[1][3];
                    }
                }
                VI_STAMP(2);
                if (!intra) {
                    // ---- next round's diagonal block and right-hand side.  The two diagonal parts and the halves of y come from
                    //      the neighbouring lanes (single-lane DPP shifts of the wave) - done while wave 1 is still updating the
                    //      blocks that hold the cross parts; only those four values wait for the mailbox.
                    const int last = M - 1;
                    const double pu00 = WaveReplay::from_prev(d[0][0]), pu01 = WaveReplay::from_prev(d[0][1]), pu11 = WaveReplay::from_prev(d[1][1]);
                    const double pv00 = WaveReplay::from_prev(d[2][2]), pv01 = WaveReplay::from_prev(d[2][3]), pv11 = WaveReplay::from_prev(d[3][3]);
                    const double nv00 = WaveReplay::from_next(d[2][2]), nv01 = WaveReplay::from_next(d[2][3]), nv11 = WaveReplay::from_next(d[3][3]);
                    const double pyu0 = WaveReplay::from_prev(yy[0]), pyu1 = WaveReplay::from_prev(yy[1]);
                    const double pyv0 = WaveReplay::from_prev(yy[2]), pyv1 = WaveReplay::from_prev(yy[3]);
                    const double nyv0 = WaveReplay::from_next(yy[2]), nyv1 = WaveReplay::from_next(yy[3]);
                    const double o00 = d[0][0], o01 = d[0][1], o11 = d[1][1], oy0 = yy[0], oy1 = yy[1];
                    const bool l0 = tid == 0, l1 = tid == 1, ll = tid == last;
                    d[0][0] = l0 ? o00 : (l1 ? pv00 : pu00);
                    d[0][1] = d[1][0] = l0 ? o01 : (l1 ? pv01 : pu01);
                    d[1][1] = l0 ? o11 : (l1 ? pv11 : pu11);
                    d[2][2] = ll ? o00 : nv00;
                    d[2][3] = d[3][2] = ll ? o01 : nv01;
                    d[3][3] = ll ? o11 : nv11;
                    yy[0] = l0 ? oy0 : (l1 ? pyv0 : pyu0);
                    yy[1] = l0 ? oy1 : (l1 ? pyv1 : pyu1);
                    yy[2] = ll ? oy0 : nyv0;
                    yy[3] = ll ? oy1 : nyv1;
                    VI_STAMP(3);
                    lds_wait<false>(sync, tgt_mb + M, bad);      // (one wave polling: no pause between polls)
                    if (tid < M) {
                        d[0][2] = d[2][0] = mb[tid];
                        d[0][3] = d[3][0] = mb[M + tid];
                        d[1][2] = d[2][1] = mb[2 * M + tid];
                        d[1][3] = d[3][1] = mb[3 * M + tid];
                    }
                }
                in_regs = true;
                VI_STAMP(4);
                if (tid == 0) lds_signal(sync + 2, 1);                // my stores of the round (diagonal block, y) are out
                if (!intra) tgt_mb += M;
                tgt_st += nw;
                ycur ^= 1;
                VI_STAMP(7);
            }
            // the sweep-end votes of jacobi_system: none rotated, or no pair of the matrix as it stands would rotate
            if (!__syncthreads_or(rotated)) { ++sweep; converged = true; break; }
            int viol = 0;
            if (tid < M) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p + 1; q < 4; ++q)
                        viol |= would_rotate(A[dg + p * M + tid], A[dg + q * M + tid], A[dg + j10(p, q) * M + tid], drop,
                                             abs_floor, conv2);
            }
            if (!__syncthreads_or(viol)) { ++sweep; converged = true; break; }
        }
    } else {
        // ================= update waves: one super-block per thread
        for (; sweep < max_sweeps; ++sweep) {
            for (int r = 0; r < m; ++r, ++nround) {
                const bool intra = r == 0;
                const double2* csc = cs0 + (nround & 1) * (4 * M);
                double b[16];
                // ---- fetch my super-block once every wave has announced its stores of the previous round (under wave 0's set-up)
                lds_wait(sync + 2, tgt_st, bad);
                if (live) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) b[e] = A[ksrc + e * nsb];
                }
                VI_STAMP(0);
                // The barrier publishes wave 0's rotations; my own outstanding LDS operations are the block loads just issued -
                // nobody else depends on them, so they stay in flight across the barrier (a plain s_barrier: __syncthreads() would
                // first wait for them) and are waited for where the update uses them.
                asm volatile("s_barrier" ::: "memory");
                VI_STAMP(1);
                // ---- S_ab <- R_a^T S_ab R_b, stored at the permuted slots; wave 1 - the blocks the next set-up waits for - first
                if (tid < 128 && !intra) __builtin_amdgcn_s_setprio(3);
                if (live) {
                    const double2 ra0 = csc[ka], ra1 = csc[M + ka];
                    const double2 rb0 = csc[kb], rb1 = csc[M + kb];
                    if (intra) {
                        rot_rows<0, 1>(b, ra0);
                        rot_rows<2, 3>(b, ra1);
                        rot_cols<0, 1>(b, rb0);
                        rot_cols<2, 3>(b, rb1);
#pragma unroll
                        for (int e = 0; e < 16; ++e) A[ksrc + e * nsb] = b[e];
                    } else {
                        const double2 ra2 = csc[2 * M + ka], ra3 = csc[3 * M + ka];
                        const double2 rb2 = csc[2 * M + kb], rb3 = csc[3 * M + kb];
                        rot_rows<0, 2>(b, ra0);
                        rot_rows<1, 3>(b, ra1);
                        rot_rows<0, 3>(b, ra2);
                        rot_rows<1, 2>(b, ra3);
                        rot_cols<0, 2>(b, rb0);
                        rot_cols<1, 3>(b, rb1);
                        rot_cols<0, 3>(b, rb2);
                        rot_cols<1, 2>(b, rb3);
                        if (serves >= 0) {
                            // the cross part of the next diagonal block of match `serves` goes to the mailbox first
                            const bool qa = q0 == 2, qb = q0 == 10;                   // (else q0 == 0)
                            mb[serves] = qa ? b[2] : (qb ? b[10] : b[0]);
                            mb[M + serves] = qa ? b[3] : (qb ? b[11] : b[1]);
                            mb[2 * M + serves] = qa ? b[6] : (qb ? b[14] : b[4]);
                            mb[3 * M + serves] = qa ? b[7] : (qb ? b[15] : b[5]);
                            lds_signal(sync, 1);
                        }
                    }
                }
                VI_STAMP(2);
                if (!intra) {
                    if (tid < 128) __builtin_amdgcn_s_setprio(0);
                    if (live) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) A[dst[e]] = b[e];
                    }
                }
                VI_STAMP(4);
                if ((tid & 63) == 0) lds_signal(sync + 2, 1);         // this wave's stores of the round are out
                tgt_st += nw;
                ycur ^= 1;
                VI_STAMP(7);
            }
            if (!__syncthreads_or(0)) { ++sweep; converged = true; break; }
            int viol = 0;
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double app = A[dg + r * M + ka];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        viol |= would_rotate(app, A[dg + c * M + kb], A[ksrc + (4 * r + c) * nsb], drop, abs_floor, conv2);
                }
            }
            if (!__syncthreads_or(viol)) { ++sweep; converged = true; break; }
        }
    }
    if (__syncthreads_or(bad)) converged = false;
    // ---- truncated solve in the eigenbasis (slot order = original order): as jacobi_system -----------------------------
    double* yc = yv + ycur * Np;
    double mx = 0.0;
    for (int i = tid; i < Np; i += NT) mx = fmax(mx, fabs(A[dg + (i & 3) * M + (i >> 2)]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mx;
    __syncthreads();
    mx = 0.0;
    for (int w = 0; w < nw; ++w) mx = fmax(mx, nd[w]);
    const double thr = rcond * mx;
    int rk = 0;
    for (int i = tid; i < Np; i += NT) {
        const double lam = A[dg + (i & 3) * M + (i >> 2)];
        const bool keep = fabs(lam) > thr;
        yc[i] = keep ? yc[i] / (lam * sc) : 0.0;
        rk += keep ? 1 : 0;
        if (lam_s && i < N) lam_s[i] = lam_raw ? lam : lam * sc;
    }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) rk += __shfl_xor(rk, o);
    if ((tid & 63) == 0) nd[tid >> 6] = (double)rk;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += nd[w];
        if (rank_s) *rank_s = (int)tot;
        if (sweeps_s) *sweeps_s = converged ? sweep : max_sweeps + 1;
        if (nround_s) *nround_s = (int)nround;
        if (round_acc) atomicAdd(round_acc, (unsigned long long)nround);
    }
    if (tid < 64) {
        constexpr int PF = 4;
        WaveReplay W;
        W.init(tid, M);
        W.load(yc, M);
        const bool has = tid < M;
        for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
            const int nb = r1 >= PF ? PF : (int)r1;
            double2 pf[PF][4];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[u][j] = make_double2(1.0, 0.0);
                if (u < nb && has) {
                    const double2* lp = logp + (r1 - 1 - u) * (int64_t)(4 * M) + tid;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[u][j] = lp[j * M];
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < nb) W.round(pf[u], ((r1 - 1 - u) % m) == 0);
        }
        W.store(yc, M);
    }
    __syncthreads();
    for (int s = tid; s < N; s += NT) Cs[s] = yc[s];
}

}  // namespace
