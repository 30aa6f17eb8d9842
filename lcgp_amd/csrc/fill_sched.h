// fill_sched.h -- filler jobs carried by the launches of the Cholesky panel chain, and their host-side scheduler.
//
// The panel chain of the factorisation (lcgp_hip.hip: leaf_fill_kernel / chain_step_kernel) is a sequence of short
// dependent launches that occupy a handful of compute units.  Every such launch can carry independent "filler" tiles
// (128 x 64 outputs, K = one outer panel) on the idle units; nothing inside a launch depends on anything else inside it,
// stream order between launches is the only ordering.  Round 1-2 carried one kind of filler (the far columns of the
// previous panel's trailing update).  Here the mechanism is general: a launch carries up to NJ job descriptors, and
// besides the trailing update the jobs of a PROGRESSIVE inverse ride along -- W = L^-1 and A^-1 = W^T W are formed panel
// by panel behind the factorisation instead of after it (the reference's per-component `eigh` + dense products,
// lcgp.py:652-654 / 704-715, become work that hides under the latency chain when a rank holds few components):
//
//   after the chain of panel P (block columns [J, pe), J = P ob):
//     TRI_T/TRI_W  W_PP   : the ob x ob block inverse from the 64 x 64 diagonal-block inverses (log2(ob) levels)
//     BROW         W[P,<P] = -W_PP T[P,<P]                         (T accumulated in V by the CUPD jobs of panels < P)
//     CUPD         T[>P,<=P] (+)= L[>P,P] W[P,<=P]                  (rank-(64 ob) update of the rows below)
//     DUPD         V[<=P,<=P] (+)= W[P,<=P]^T W[P,<=P]              (V ends as A^-1; rows of panel P are first written here)
//
// This header is plain C++ (no HIP): the descriptors are shared with the device code, and the scheduler is exercised on
// the CPU by tests/test_fill_sched.py (through tests/native/dump_plan.cpp, which prints a plan as text): the plan is
// replayed on numpy matrices with the semantics of the kernels and every read-after-write and write-after-write constraint
// is checked launch by launch -- and, for the persistent form, segment by segment against the derived dependency lists.
#ifndef LCGP_FILL_SCHED_H
#define LCGP_FILL_SCHED_H

#include <stddef.h>
#include <string.h>
#include <utility>
#include <vector>

namespace lcgp_fill {

enum FillType { FILL_NONE = 0, FILL_SYRK = 1, FILL_BROW = 2, FILL_CUPD = 3, FILL_DUPD = 4, FILL_TRI_T = 5, FILL_TRI_W = 6 };

constexpr int NJ = 6;      // job descriptors per launch

// Output tiles are 128 rows x 64 columns: R counts 128-row blocks, j 64-column blocks, K ranges are in 64-blocks
// [kb0, kb1) (block columns of L for SYRK / CUPD, block rows of W for BROW / CUPD / DUPD).  Tile enumeration t:
//   SYRK : column-major over j in [j0, j1), rows R in [j/2, R1)            (the lower trapezoid right of column j0)
//   BROW, CUPD : row-major over R in [R0, R1), j in [j0, j1)
//   DUPD : row-major over R >= 0, j in [0, 2R + 2):  t = R (R + 1) + j       (lower triangle incl. whole diagonal blocks)
//   TRI_T, TRI_W : the 64 x 64 tiles of one level of the block inverse: R0 = level size mb (64-blocks), R1 = pairs,
//                  j0 = first pair; tiles = pairs * mb * mb
struct FillJob {
    int type;
    int nblk;            // blocks of this job in this launch = tiles x components (component = fastest index)
    int t0;              // first tile
    int R0, R1, j0, j1;
    int kb0, kb1;
};

struct FillSet {
    void* M; void* W; void* V;     // component-0 bases of L, L^-1, scratch / A^-1
    size_t mat;                     // elements per component matrix
    int npad, nb, q;
    int njobs;
    int nblk;                       // blocks of all jobs
    FillJob job[NJ];
};

inline long syrk_tiles(int R1, int j0, int j1) {
    long n = 0;
    for (int j = j0; j < j1; ++j) n += R1 - (j >> 1);
    return n;
}
inline long dupd_tiles(int R1) { return (long)R1 * (R1 + 1); }     // rows [0, R1)

// ---------------------------------------------------------------------------------------------------
// scheduler
// ---------------------------------------------------------------------------------------------------
struct QJob {
    FillJob j = {};          // type and ranges (nblk / t0 are filled per launch)
    long total = 0;          // tiles
    long next = 0;           // tiles handed out so far
    long avail = 0;          // tiles completed in EARLIER launches (what a dependent job may rely on)
    int ready_launch = 0;    // first launch index in which the job may run (inputs produced by the chain)
    int dep[2] = {-1, -1};   // jobs that must be complete before this one starts
    int wave = -1;           // wavefront predecessor: the same tiles one panel earlier (CUPD / DUPD); BROW: the CUPD
                             // of the previous panel, whose first rows are this panel's T
    int ncols = 1;           // CUPD: tiles per row
    bool small = false;      // may ride on a launch that has no special workgroup (the few 64 x 64 tiles of a block inverse)
    bool complete() const { return next >= total; }
};

class FillQueue {
 public:
    std::vector<QJob> jobs;
    int launch = 0;          // index of the launch being assembled
    int q = 1;
    int syrk_job = -1;       // the trailing-update job with a deadline at the end of the current panel chain

    int add(const QJob& jb) { jobs.push_back(jb); return (int)jobs.size() - 1; }

    bool done(int id) const { return id < 0 || jobs[id].avail >= jobs[id].total; }

    // first tile index the job may NOT touch in the launch being assembled
    long limit(const QJob& jb) const {
        if (launch < jb.ready_launch || !done(jb.dep[0]) || !done(jb.dep[1])) return jb.next;
        if (jb.wave < 0) return jb.total;
        const QJob& pv = jobs[jb.wave];
        if (pv.avail >= pv.total) return jb.total;
        if (jb.j.type == FILL_BROW) {
            // T[P, <P] is complete when the previous panel's CUPD has finished this panel's rows (its first rows; the
            // wavefront below makes that imply the same of every older CUPD)
            const long need = (long)(jb.j.R1 - jb.j.R0) * pv.ncols;
            return pv.avail >= need ? jb.total : jb.next;
        }
        if (jb.j.type == FILL_CUPD) {
            // rows the predecessor has completed: [pv.R0, pv.R0 + full); this job starts at its own R0 > pv.R0
            const long full = pv.avail / pv.ncols;
            long rows = pv.j.R0 + full - jb.j.R0;
            if (rows < 0) rows = 0;
            const long lim = rows * jb.ncols;
            return lim < jb.total ? lim : jb.total;
        }
        // DUPD: the same row-major enumeration; the predecessor's tiles are a prefix of this job's
        return pv.avail < jb.total ? pv.avail : jb.total;
    }

    bool pending() const {
        for (const QJob& jb : jobs) if (!jb.complete()) return true;
        return false;
    }

    // next 128-row block a CUPD job would work on (for nearest-row-first selection)
    static long cupd_row(const QJob& jb) { return jb.j.R0 + jb.next / jb.ncols; }

    // Fills `fs.job[]` with up to `cap_blocks` blocks of ready work (NJ descriptors at most) and returns the block count.
    // allow_big = false: only the small block-inverse jobs (the launch has no long-running workgroup to hide behind).
    // urgent_row: CUPD tiles of rows below this 128-row block feed the next BROW and go before the trailing update.
    // Order: block inverse, row of the inverse, urgent CUPD rows, trailing update (deadline: end of the panel chain),
    // other CUPD rows nearest first, DUPD oldest first.
    int take(long cap_blocks, bool allow_big, int urgent_row, FillSet& fs, bool with_dupd = true) {
        fs.njobs = 0;
        fs.nblk = 0;
        auto cost = [&](const QJob&) { return (long)q; };
        auto emit = [&](QJob& jb, long n) {
            if (n <= 0 || fs.njobs >= NJ) return;
            FillJob& o = fs.job[fs.njobs++];
            o = jb.j;
            o.t0 = (int)jb.next;
            o.nblk = (int)(n * q);
            jb.next += n;
            fs.nblk += o.nblk;
            cap_blocks -= n * cost(jb);
        };
        auto room = [&](const QJob& jb) { return cap_blocks / cost(jb); };
        for (QJob& jb : jobs) {                                   // 1, 2: block inverse and BROW
            if (jb.complete() || (jb.j.type != FILL_TRI_T && jb.j.type != FILL_TRI_W && jb.j.type != FILL_BROW)) continue;
            if (!allow_big && !jb.small) continue;
            long n = limit(jb) - jb.next;
            if (n > room(jb)) n = room(jb);
            emit(jb, n);
        }
        if (!allow_big) return fs.nblk;
        auto cupd_pass = [&](long row_end) {                      // nearest rows first, one row block per pick
            for (;;) {
                int best = -1;
                long brow = row_end;
                for (int i = 0; i < (int)jobs.size(); ++i) {
                    QJob& jb = jobs[i];
                    if (jb.j.type != FILL_CUPD || jb.complete() || limit(jb) <= jb.next) continue;
                    const long r = cupd_row(jb);
                    if (r < brow) { brow = r; best = i; }
                }
                if (best < 0 || cap_blocks < q || fs.njobs >= NJ) break;
                QJob& jb = jobs[best];
                // up to the end of the row block group the job is in (whole rows of the next two row blocks)
                long row_stop = (brow / 2 + 1) * 2;
                if (row_stop > row_end) row_stop = row_end;
                long n = (row_stop - jb.j.R0) * jb.ncols - jb.next;
                const long lim = limit(jb) - jb.next;
                if (n > lim) n = lim;
                if (n > room(jb)) n = room(jb);
                // the same job may be picked again in this launch: merge with its previous descriptor
                if (fs.njobs > 0 && fs.job[fs.njobs - 1].type == FILL_CUPD && fs.job[fs.njobs - 1].kb0 == jb.j.kb0 &&
                    fs.job[fs.njobs - 1].t0 + fs.job[fs.njobs - 1].nblk / q == jb.next) {
                    fs.job[fs.njobs - 1].nblk += (int)(n * q);
                    jb.next += n;
                    fs.nblk += (int)(n * q);
                    cap_blocks -= n * cost(jb);
                } else {
                    emit(jb, n);
                }
                if (n <= 0) break;
            }
        };
        cupd_pass(urgent_row);                                    // 3
        if (syrk_job >= 0 && !jobs[syrk_job].complete()) {        // 4
            QJob& jb = jobs[syrk_job];
            long n = limit(jb) - jb.next;
            if (n > room(jb)) n = room(jb);
            emit(jb, n);
        }
        cupd_pass(1L << 40);                                      // 5
        if (!with_dupd) return fs.nblk;
        for (QJob& jb : jobs) {                                   // 6
            if (jb.j.type != FILL_DUPD || jb.complete()) continue;
            long n = limit(jb) - jb.next;
            if (n > room(jb)) n = room(jb);
            emit(jb, n);
        }
        return fs.nblk;
    }

    // Final flush, after the last launch of the chain: as take() without a capacity, and the pending DUPD jobs merged
    // into K bands (a tile that still lacks the panels P .. last receives them in ONE visit with a long K loop, the
    // shape of the one-launch A^-1 = W^T W of the non-progressive path).  A band needs every BROW of its K range complete.
    int take_final(FillSet& fs, int kb_end) {
        int first_d = -1, last_d = -1;
        bool brows_done = true;
        for (int i = 0; i < (int)jobs.size(); ++i) {
            const QJob& jb = jobs[i];
            if ((jb.j.type == FILL_BROW || jb.j.type == FILL_TRI_T || jb.j.type == FILL_TRI_W || jb.j.type == FILL_CUPD) &&
                jb.avail < jb.total)
                brows_done = false;
            if (jb.j.type == FILL_DUPD && !jb.complete()) {
                if (first_d < 0) first_d = i;
                last_d = i;
            }
        }
        // until every row of W is final only the jobs that lead there run (the A^-1 updates wait for the merged launch:
        // one visit per tile with a long K loop instead of one read-modify-write pass per panel)
        if (!brows_done) return take(1L << 40, true, 1 << 30, fs, false);
        if (first_d < 0) return take(1L << 40, true, 1 << 30, fs);
        // every row of W is final: merged bands, oldest pending panel first.  Job i still lacks its tiles [next_i, ...);
        // the tiles [next_i, hi) -- hi = the next older job's `next`, or the last tile of all for the oldest -- lack
        // exactly the panels i .. last, so they form ONE descriptor with the K range [kb0_i, kb_end).
        int bands = 0;
        long hi = jobs[last_d].total;
        for (int i = first_d; i <= last_d; ++i) {
            if (jobs[i].j.type != FILL_DUPD) continue;
            const long lo = jobs[i].next < hi ? jobs[i].next : hi;
            if (hi > lo) ++bands;
            hi = lo;
        }
        if (bands > NJ) return take(1L << 40, true, 1 << 30, fs);      // (more bands than descriptors: plain order)
        fs.njobs = 0;
        fs.nblk = 0;
        hi = jobs[last_d].total;
        for (int i = first_d; i <= last_d; ++i) {
            QJob& jb = jobs[i];
            if (jb.j.type != FILL_DUPD) continue;
            const long lo = jb.next < hi ? jb.next : hi;
            if (hi > lo) {
                FillJob& o = fs.job[fs.njobs++];
                o = jb.j;
                o.kb1 = kb_end;
                o.t0 = (int)lo;
                o.nblk = (int)((hi - lo) * q);
                fs.nblk += o.nblk;
            }
            hi = lo;
            jb.next = jb.total;
        }
        return fs.nblk;
    }

    void end_launch() {
        for (QJob& jb : jobs) jb.avail = jb.next;
        ++launch;
    }
};

// ---------------------------------------------------------------------------------------------------
// The launch plan of one factorisation (+ progressive inverse): computed on the host before anything is enqueued, from
// the block count, the number of components and the schedule parameters alone.  lcgp_hip.hip executes it launch by
// launch; tests/test_fill_sched.py replays it on the CPU (tests/native/dump_plan.cpp prints it).
// ---------------------------------------------------------------------------------------------------
struct PlanParams {
    int nb;                  // 64-blocks per side (even)
    int q;                   // components of the rank
    int ob;                  // outer panel width in 64-blocks
    int syrk_small_tiles, fill_leaf, fill_step, leaf_in_wide;     // lcgp_sched fields of the same names
    bool progressive;        // queue the jobs of the progressive inverse
    bool far_rides;          // the far columns of a trailing update ride on the next panel's chain (else: one wide launch)
    bool with_dupd = true;   // progressive: A^-1 = W^T W is accumulated behind the chain too (else only L^-1 is; the caller
                             // then forms A^-1 in one launch after the factorisation)
    int pair_tiles = 0;      // > 0: two consecutive panels share ONE trailing update with K = 2 ob on the columns between the second
                             // panel and the far columns when that region has at least this many 64x64 tiles x components (the
                             // first panel only updates the second panel's own columns); 0 = every panel updates everything
};

enum LaunchKind {
    L_LEAF = 1,              // diagonal block J (+ filler)
    L_STEP = 2,              // chain step of block column c (+ filler)
    L_TRAIL = 3,             // wide trailing update of panel [J, pe) on the block columns [c_lo, c_hi)
    L_FILL = 4               // filler jobs on their own
};

struct Launch {
    // every byte zero, padding and the executor-owned pointers of `fs` included: a plan is compared and hashed as bytes
    Launch() { memset((void*)this, 0, sizeof(*this)); }
    int kind = 0;
    int J = 0, pe = 0, c = 0;
    int diag_end = 0, has_special = 0, n_trmm = 0, n_upd = 0;       // L_STEP
    int c_lo = 0, c_hi = 0, tiles128 = 0, with_leaf = 0;            // L_TRAIL
    FillSet fs;                                                     // job descriptors (base pointers are set by the executor)
};

inline long rect_tiles(int nb, int c_lo, int c_hi) { return syrk_tiles(nb / 2, c_lo, c_hi); }
inline int trapezoid_tiles(int nb, int c_lo, int c_hi) { return (c_hi - c_lo) * nb - (c_lo + c_hi - 1) * (c_hi - c_lo) / 2; }

class Planner {
 public:
    explicit Planner(const PlanParams& p) : pp(p) { fq.q = p.q; }
    std::vector<Launch> launches;
    bool inverse_planned = false;

    void run() {
        const int nb = pp.nb, ob = pp.ob, q = pp.q;
        const bool t128 = (ob & 1) == 0;
        inverse_planned = pp.progressive;
        bool leaf_done = false;
        // filler capacity of a panel's chain launches, with and without a diagonal-block launch of its own
        const int cap_with_leaf = pp.fill_leaf + (ob - 1) * pp.fill_step;
        const int cap_no_leaf = (ob - 1) * pp.fill_step;
        for (int J = 0; J < nb; J += ob) {
            const int pe = J + ob < nb ? J + ob : nb;
            const int mid = pe + ob < nb ? pe + ob : nb;     // the next panel's own columns are never filler
            auto first_filler_column = [&](int cap_blocks) {
                int c = nb;
                // filler rides on the chain launches of the NEXT panel [pe, mid): it must stay clear of that panel's
                // columns and, when those chain steps pre-apply the panel to the diagonal block (mid, mid) for a
                // trailing-update launch that factors it, of column `mid` too
                const int lo = mid + (pp.leaf_in_wide ? 2 : 0);
                if (t128 && cap_blocks >= q && lo < nb) {
                    const long cap_tiles = cap_blocks / q;
                    while (c - 2 >= lo && rect_tiles(nb, c - 2, nb) <= cap_tiles) c -= 2;
                }
                return c;
            };
            auto wide128 = [&](int c_hi) {
                return t128 && (long long)q * trapezoid_tiles(nb / 2, pe / 2, c_hi / 2) >= pp.syrk_small_tiles;
            };
            // decided BEFORE the panel's chain, which then pre-applies the panel to the next diagonal block:
            // when the update runs on 64x64 tiles (few tiles: late panels, few components) it also factors the next
            // panel's first diagonal block.  With many 128x128 tiles that does not pay: the diagonal-block launch
            // carries filler of its own and the 8-wave tile kernel is the faster one.
            int cf = nb;                                       // first filler column
            bool next_leaf = false;
            if (pe < nb) {
                cf = first_filler_column(pp.far_rides ? cap_with_leaf : 0);
                if (pp.leaf_in_wide && !wide128(cf)) {
                    const int cf3 = first_filler_column(pp.far_rides ? cap_no_leaf : 0);
                    // (that kernel holds two workgroups per CU, the plain 64-tile kernel four: launches of few rounds)
                    if (!wide128(cf3) && (long long)q * trapezoid_tiles(nb, pe, cf3) <= pp.leaf_in_wide) {
                        next_leaf = true;
                        cf = cf3;
                    }
                }
            }
            // Paired panels: the update of the columns [mid, cf) by THIS panel waits for the next one and runs with K = 2 ob --
            // half the read-modify-write passes over that region, twice the K loop per tile (measured 6-9 % faster on it:
            // profiles/r06_deferred_k512_microbench.txt).  The far columns keep riding on the chain launches panel by panel.
            int k_lo = J;                                      // first block column of the K range of this panel's wide update
            int c_hi = cf;
            if (pair_J >= 0) {                                 // second panel of a pair
                cf = pair_cf;
                next_leaf = false;                             // (the diagonal block (pe, pe) still lacks the first panel)
                k_lo = pair_J;
                c_hi = cf;
                pair_J = -1;
            } else if (pp.pair_tiles > 0 && !pp.progressive && !next_leaf && pe - J == ob && mid - pe == ob && mid + ob + 2 <= cf &&
                       (long long)q * trapezoid_tiles(nb, mid, cf) >= pp.pair_tiles) {
                pair_J = J;
                pair_cf = cf;
                next_leaf = false;
                c_hi = mid;                                    // only the next panel's own columns now
            }
            panel(J, pe, leaf_done, next_leaf);
            if (pe >= nb) break;
            if (pe < c_hi) {
                Launch l;
                l.kind = L_TRAIL; l.J = k_lo; l.pe = pe; l.c_lo = pe; l.c_hi = c_hi;
                l.tiles128 = t128 && (long long)q * trapezoid_tiles(nb / 2, pe / 2, c_hi / 2) >= pp.syrk_small_tiles ? 1 : 0;
                l.with_leaf = next_leaf ? 1 : 0;
                l.fs.njobs = 0; l.fs.nblk = 0;
                launches.push_back(l);
                fq.end_launch();
            }
            leaf_done = next_leaf && pe < c_hi;
            queue_far_update(J, pe, cf);
        }
        // the tail of the progressive inverse: what the chain launches did not carry, in as few dependent launches as
        // the job dependencies allow; the remaining A^-1 updates merged into K bands
        int guard = 0;
        while (fq.pending()) {
            Launch l;
            l.kind = L_FILL;
            fq.take_final(l.fs, kb_end);
            if (l.fs.nblk <= 0 || ++guard > 4 * nb + 16) { failed = true; return; }
            launches.push_back(l);
            fq.end_launch();
        }
    }
    bool failed = false;

 private:
    int pair_J = -1, pair_cf = 0;       // first panel of an open pair and the far boundary the two share
    PlanParams pp;
    FillQueue fq;
    int urgent_row = 0;           // CUPD rows below this 128-row block feed the next row of the inverse
    int last_dupd = -1, last_cupd = -1;
    int kb_end = 0;               // end of the last panel whose inverse jobs are queued

    // the far columns [cf, nb) of the trailing update of panel [J, pe) become the filler job with a deadline at the
    // end of the next panel's chain
    void queue_far_update(int J, int pe, int cf) {
        fq.syrk_job = -1;
        if (cf >= pp.nb) return;
        QJob jb;
        jb.j.type = FILL_SYRK;
        jb.j.R0 = 0; jb.j.R1 = pp.nb / 2; jb.j.j0 = cf; jb.j.j1 = pp.nb; jb.j.kb0 = J; jb.j.kb1 = pe;
        jb.total = syrk_tiles(pp.nb / 2, cf, pp.nb);
        jb.ready_launch = fq.launch;
        fq.syrk_job = fq.add(jb);
    }

    // Jobs of the progressive inverse for the panel [J, pe), queued just before the launch that finishes the panel's
    // last block column (`last_step`: that launch exists, i.e. there are rows below the panel).
    void queue_inverse_jobs(int J, int pe, bool last_step) {
        const int nb = pp.nb;
        const int now = fq.launch;                 // the diagonal blocks of the panel are final before this launch
        int prev = -1;
        for (int mb = 1; mb < pp.ob; mb *= 2) {    // block inverse of the panel, level by level (T, then W)
            const int pair0 = J / (2 * mb);
            int npair = 0;
            for (int pr = pair0; 2 * pr * mb + mb < pe; ++pr) ++npair;      // pairs whose second half exists
            if (npair == 0) break;
            for (int step = 0; step < 2; ++step) {
                QJob jb;
                jb.j.type = step == 0 ? FILL_TRI_T : FILL_TRI_W;
                jb.j.R0 = mb; jb.j.R1 = npair; jb.j.j0 = pair0; jb.j.j1 = 0; jb.j.kb0 = J; jb.j.kb1 = pe;
                jb.total = (long)npair * mb * mb;
                jb.ready_launch = now;
                jb.dep[0] = prev;
                jb.small = true;
                prev = fq.add(jb);
            }
        }
        const int inv_done = prev;
        int brow = -1;
        if (J > 0) {
            QJob jb;
            jb.j.type = FILL_BROW;
            jb.j.R0 = J / 2; jb.j.R1 = (pe + 1) / 2; jb.j.j0 = 0; jb.j.j1 = J; jb.j.kb0 = J; jb.j.kb1 = pe;
            jb.ncols = J;
            jb.total = (long)(jb.j.R1 - jb.j.R0) * J;
            jb.ready_launch = now;
            jb.dep[0] = inv_done;
            jb.wave = last_cupd;                   // T[P, <P] is complete once that job has finished this panel's rows
            brow = fq.add(jb);
        }
        const int w_done = brow >= 0 ? brow : inv_done;
        if (pe < nb) {
            QJob jb;
            jb.j.type = FILL_CUPD;
            jb.j.R0 = pe / 2; jb.j.R1 = nb / 2; jb.j.j0 = 0; jb.j.j1 = pe; jb.j.kb0 = J; jb.j.kb1 = pe;
            jb.ncols = pe;
            jb.total = (long)(jb.j.R1 - jb.j.R0) * jb.ncols;
            jb.ready_launch = now + (last_step ? 1 : 0);   // the panel's last block column of L is final after that launch
            jb.dep[0] = w_done;
            jb.wave = last_cupd;
            last_cupd = fq.add(jb);
        }
        if (pp.with_dupd) {
            QJob jb;
            jb.j.type = FILL_DUPD;
            jb.j.R0 = 0; jb.j.R1 = (pe + 1) / 2; jb.j.j0 = 0; jb.j.j1 = pe; jb.j.kb0 = J; jb.j.kb1 = pe;
            jb.total = dupd_tiles(jb.j.R1);
            jb.ready_launch = now;
            jb.dep[0] = w_done;
            jb.wave = last_dupd;
            last_dupd = fq.add(jb);
        }
        kb_end = pe;
        const int nxt = pe + pp.ob < nb ? pe + pp.ob : nb;
        urgent_row = (nxt + 1) / 2;
    }

    // One outer panel [J, pe): the diagonal block J on its own (unless the previous trailing-update launch factored
    // it), then ONE launch per 64-column step; every launch may carry filler jobs.
    void panel(int J, int pe, bool leaf_done, bool next_leaf_in_wide) {
        const int nb = pp.nb;
        if (!leaf_done) {
            Launch l;
            l.kind = L_LEAF; l.J = J; l.pe = pe;
            fq.take(pp.fill_leaf, true, urgent_row, l.fs);
            launches.push_back(l);
            fq.end_launch();
        }
        bool queued = false;
        for (int c = J; c < pe && c + 1 < nb; ++c) {
            Launch l;
            l.kind = L_STEP; l.J = J; l.pe = pe; l.c = c;
            l.diag_end = pe + (next_leaf_in_wide ? 1 : 0);
            l.has_special = c + 1 < pe ? 1 : 0;
            l.n_trmm = nb - 1 - c;
            l.n_upd = 0;
            if (c > J && c + 1 < pe) l.n_upd = nb - (c + 1) - 1;     // the tiles below the diagonal of column c + 1
            // the launch of the panel's last block column: every diagonal block of the panel is final before it
            if (pp.progressive && !queued && c == pe - 1) { queue_inverse_jobs(J, pe, true); queued = true; }
            // a step that ends in a diagonal block lasts as long as a filler tile; the last step of a panel is short
            // and only takes the few 64x64 tiles of the block inverse along
            fq.take(pp.fill_step, l.has_special != 0, urgent_row, l.fs);
            launches.push_back(l);
            fq.end_launch();
        }
        if (pp.progressive && !queued) queue_inverse_jobs(J, pe, false);       // the last panel: no rows below
        // what is left of the trailing update the chain could not carry runs as one plain launch
        if (fq.syrk_job >= 0 && !fq.jobs[fq.syrk_job].complete()) {
            QJob& jb = fq.jobs[fq.syrk_job];
            Launch l;
            l.kind = L_FILL;
            l.fs.njobs = 1;
            l.fs.job[0] = jb.j;
            l.fs.job[0].t0 = (int)jb.next;
            l.fs.job[0].nblk = (int)((jb.total - jb.next) * pp.q);
            l.fs.nblk = l.fs.job[0].nblk;
            jb.next = jb.total;
            launches.push_back(l);
            fq.end_launch();
        }
        fq.syrk_job = -1;
    }
};

}  // namespace lcgp_fill

#endif
