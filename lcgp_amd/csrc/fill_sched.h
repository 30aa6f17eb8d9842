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
    int wide;            // SYRK / CUPD: 1 = 128 x 128 output tiles -- the columns go in PAIRS (j, j + 1), j even (j0, j1, kb0
                         // even), same enumeration over the pairs; such a tile is twice the work of a 128 x 64 one at half the
                         // operand traffic per flop and runs at the rate of the 128-tile kernel
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
inline long syrk_tiles_wide(int R1, int j0, int j1) {             // column pairs (j0, j1 even)
    long n = 0;
    for (int j = j0; j < j1; j += 2) n += R1 - (j >> 1);
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
        // (the capacity counts 128 x 64 tiles: a wide tile takes two units)
        auto cost = [&](const QJob& jb) { return (long)q * (jb.j.wide ? 2 : 1); };
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
    bool interleaved = false; // order for the persistent launch (Planner::run_interleaved): near / far trailing updates,
                              // the far part cut into chunks that alternate with the next panel's chain
    bool with_trtri = false;  // interleaved: the triangular inverse W = L^-1 (level-parallel products) is part of the same
                              // sequence -- its early levels fill the chain-bound end of the factorisation
    int trtri_all_small = 0;  // ... every level on 64 x 64 tiles (small problems); else only the first
    int tri_fill_from = 256;  // ... they start to ride when a chain step's share of the far update drops below this many blocks
    bool psolve = true;       // ... and the rows below the chain rows are solved per panel with the panel's 256 x 256 inverse
    bool fill_wide = false;   // launch-by-launch plan: the far columns of the trailing update and the rank-(64 ob) updates of the
                              // progressive inverse ride as 128 x 128 tiles (FillJob::wide; needs an even ob)
};

enum LaunchKind {
    L_LEAF = 1,              // diagonal block J (+ filler)
    L_STEP = 2,              // chain step of block column c (+ filler)
    L_TRAIL = 3,             // wide trailing update of panel [J, pe) on the block columns [c_lo, c_hi)
    L_FILL = 4,              // filler jobs on their own
    L_TRI = 5,               // one step (T = L21 W11 or W21 = -W22 T) of one level of the triangular inverse, a range of pairs
    L_PSOLVE = 6             // panel solve of the rows below the chain rows: L[R, jt] = sum_{kt <= jt} A[R, kt] W_PP[jt, kt]^T for the
                             // 128-column tile jt (= c_lo) of the panel [J, pe) and the 128-row tiles from block row r_lo on
};

struct Launch {
    // every byte zero, padding and the executor-owned pointers of `fs` included: a plan is compared and hashed as bytes
    Launch() { memset((void*)this, 0, sizeof(*this)); }
    int kind = 0;
    int J = 0, pe = 0, c = 0;
    int diag_end = 0, has_special = 0, n_trmm = 0, n_upd = 0;       // L_STEP
    int trmm_r0 = 0, upd_r0 = 0;                                    // L_STEP: first block row of its n_trmm solve tiles / n_upd delayed
                                                                    // update tiles (0 = c + 1 / c + 2: all rows below the diagonal)
    int c_lo = 0, c_hi = 0, tiles128 = 0, with_leaf = 0;            // L_TRAIL
    int t_first = 0, t_count = 0;                                   // L_TRAIL: a sub-range of its tiles (t_count = 0: all of them)
    int r_lo = 0, r_hi = 0;                                         // L_TRAIL: block rows [max(column, r_lo), r_hi) of every block
                                                                    // column (r_hi = 0: down to the last row)
    int tri_mb = 0, tri_p0 = 0, tri_np = 0, tri_w = 0;              // L_TRI: level block size (tiles), first pair, pairs, 0 = T / 1 = W
                                                                    // (tiles128 = tile size)
    FillSet fs;                                                     // job descriptors (base pointers are set by the executor)
};

inline long rect_tiles(int nb, int c_lo, int c_hi) { return syrk_tiles(nb / 2, c_lo, c_hi); }
// tiles of a trailing-update region: tile columns [c_lo, c_hi), in column c the tile rows [max(c, r_lo), r_hi)
inline long region_tiles(int c_lo, int c_hi, int r_lo, int r_hi) {
    long n = 0;
    for (int c = c_lo; c < c_hi; ++c) {
        const int a = c > r_lo ? c : r_lo;
        if (r_hi > a) n += r_hi - a;
    }
    return n;
}
inline int trapezoid_tiles(int nb, int c_lo, int c_hi) { return (c_hi - c_lo) * nb - (c_lo + c_hi - 1) * (c_hi - c_lo) / 2; }

class Planner {
 public:
    explicit Planner(const PlanParams& p) : pp(p) { fq.q = p.q; }
    std::vector<Launch> launches;
    bool inverse_planned = false;

    // Order for the persistent launch (dag_kernel takes the tasks of all segments in sequence): the trailing update of
    // panel P is cut into its NEAR part -- the block columns of panel P + 1, all the next chain reads or writes -- and its
    // FAR part, and the far part into chunks that alternate with the launches of the chain of panel P + 1:
    //     ... near(P) | leaf | far(P) 1/5 | step | far(P) 2/5 | step | ... | near(P+1) | ...
    // A chain task then never waits for far tiles (the dependencies are derived per segment from the blocks it touches),
    // the workgroups that are not on the chain always find update tiles to run, and no update ends in a half-empty round
    // of tiles because nothing ends at all: the next segment's tiles follow.  Executed launch by launch the list is valid
    // too (it is what tests/test_fill_sched.py replays), only slow.
    void run_interleaved() {
        const int nb = pp.nb, ob = pp.ob, q = pp.q;
        const bool t128 = (ob & 1) == 0;
        // pending "fill" work of the previous panel: the rest of its near columns and its far columns (in chunks)
        struct Fill { bool on; Launch l; long ntiles; long given; };
        Fill rest = {false, Launch(), 0, 0}, far = {false, Launch(), 0, 0};
        auto emit_chunk = [&](Fill& f, long upto) {
            if (!f.on || upto > f.ntiles) upto = f.on ? f.ntiles : 0;
            if (!f.on || upto <= f.given) return;
            Launch l = f.l;
            l.t_first = (int)f.given;
            l.t_count = (int)(upto - f.given);
            launches.push_back(l);
            f.given = upto;
        };
        std::vector<Launch> bulk;        // bulk parts of this panel's chain steps, waiting for a slot in the order
        // Units of the triangular inverse (one step of one level for a run of pairs), sorted by the panel that completes
        // their rows of L, then by level: every unit comes behind the units it reads from.
        struct Tri { Launch l; int ready; long blocks; };
        std::vector<Tri> tri;
        if (pp.with_trtri) {
            for (int P = 0; P * ob < nb; ++P) {
                for (int mb64 = 1; mb64 < nb; mb64 *= 2) {
                    const int tm128 = (mb64 == 1 || pp.trtri_all_small || !t128) ? 0 : 1;
                    const int u = tm128 ? 2 : 1, mb = mb64 / u, nbt = nb / u;
                    for (int w = 0; w < 2; ++w) {
                        int first = -1, count = 0;
                        auto flush = [&]() {
                            if (count == 0) return;
                            Tri t;
                            t.l = Launch();
                            t.l.kind = L_TRI; t.l.tiles128 = tm128; t.l.tri_mb = mb; t.l.tri_p0 = first; t.l.tri_np = count; t.l.tri_w = w;
                            t.l.fs.njobs = 0; t.l.fs.nblk = 0;
                            t.ready = P;
                            t.blocks = (long)count * mb * mb * q;
                            tri.push_back(t);
                            count = 0; first = -1;
                        };
                        for (int pr = 0; (2 * pr + 1) * mb < nbt; ++pr) {
                            int last = (2 * pr + 2) * mb * u;            // one past the last 64-block row the pair reads
                            if (last > nb) last = nb;
                            const int rp = (last - 1) / ob;
                            if (rp == P) { if (count == 0) first = pr; ++count; }
                            else flush();
                        }
                        flush();
                    }
                }
            }
        }
        // Rows below the chain rows: with 128-aligned panels and the inverse in the sequence they skip the 64-column steps
        // altogether -- once the panel's diagonal block and ITS inverse W_PP exist (the first levels of the triangular
        // inverse, which ride right behind the panel's chain), L[R, panel] = A[R, panel] W_PP^T is ONE product on 128 x 128
        // tiles per 128 columns (a quarter of the arithmetic of the four solve + update steps, at the tile kernel's rate).
        const bool use_psolve = pp.with_trtri && t128 && !pp.trtri_all_small && pp.psolve;
        size_t tri_next = 0;
        int panel_idx = 0;
        auto emit_tri = [&](long budget_blocks, int done_panel) {
            // units whose rows of L are complete (panels <= done_panel), in order, up to the budget
            while (tri_next < tri.size() && tri[tri_next].ready <= done_panel && budget_blocks > 0) {
                launches.push_back(tri[tri_next].l);
                budget_blocks -= tri[tri_next].blocks;
                ++tri_next;
            }
        };
        for (int J = 0; J < nb; J += ob, ++panel_idx) {
            const int pe = J + ob < nb ? J + ob : nb;
            const int ne = pe + ob < nb ? pe + ob : nb;            // end of the next panel
            const int rch = ne + ob < nb ? ne + ob : nb;           // chain rows: this panel's, the next one's and the one after
            {
                Launch l;
                l.kind = L_LEAF; l.J = J; l.pe = pe;
                l.fs.njobs = 0; l.fs.nblk = 0;
                launches.push_back(l);
            }
            emit_chunk(rest, rest.ntiles);                          // (needed by the bulk rows of this panel's chain)
            int nsteps = 0;
            for (int c = J; c < pe && c + 1 < nb; ++c) ++nsteps;
            int si = 0;
            for (int c = J; c < pe && c + 1 < nb; ++c, ++si) {
                // chain part: the rows the following diagonal blocks need; bulk part: all rows below them
                const bool upd = c > J && c + 1 < pe;
                Launch l;
                l.kind = L_STEP; l.J = J; l.pe = pe; l.c = c;
                l.diag_end = pe;
                l.fs.njobs = 0; l.fs.nblk = 0;
                Launch ch = l, bk = l;
                ch.has_special = c + 1 < pe ? 1 : 0;
                ch.trmm_r0 = c + 1; ch.n_trmm = (rch < nb ? rch : nb) - (c + 1);
                ch.upd_r0 = c + 2; ch.n_upd = upd ? (rch < nb ? rch : nb) - (c + 2) : 0;
                if (ch.n_upd < 0) ch.n_upd = 0;
                bk.has_special = 0;
                bk.trmm_r0 = rch; bk.n_trmm = use_psolve ? 0 : nb - rch;
                bk.upd_r0 = rch; bk.n_upd = (upd && !use_psolve) ? nb - rch : 0;
                launches.push_back(ch);
                // fill behind the chain step: the bulk rows of the step before, a share of the far update
                if (!bulk.empty()) { launches.push_back(bulk.back()); bulk.pop_back(); }
                const long far_before = far.on ? far.given : 0;
                emit_chunk(far, far.ntiles * (si + 1) / (nsteps > 0 ? nsteps : 1));
                // the chain-bound end of the factorisation: units of the inverse whose rows of L were complete TWO panels ago
                // (their last producers are well behind in the sequence) take the room the far update no longer fills
                {
                    const long far_blocks = ((far.on ? far.given : 0) - far_before) * (far.on && far.l.tiles128 ? 4 : 1) * q;
                    if (far_blocks < pp.tri_fill_from) emit_tri(pp.tri_fill_from - far_blocks, panel_idx - 2);
                }
                if (bk.n_trmm > 0) bulk.push_back(bk);
            }
            emit_chunk(far, far.ntiles);
            far.on = false; rest.on = false;
            if (pe < nb) {
                // the near update in two parts: what the next chain reads (on 64 x 64 tiles: it is on the critical path) ...
                Launch l;
                l.kind = L_TRAIL; l.J = J; l.pe = pe; l.c_lo = pe; l.c_hi = ne; l.tiles128 = 0; l.with_leaf = 0;
                l.r_lo = 0; l.r_hi = rch;
                l.t_first = 0; l.t_count = (int)region_tiles(pe, ne, 0, rch);
                l.fs.njobs = 0; l.fs.nblk = 0;
                launches.push_back(l);
            }
            while (!bulk.empty()) { launches.push_back(bulk.back()); bulk.pop_back(); }
            if (use_psolve) {
                // the panel's own levels of the inverse (the 64-block pairs inside it, then its 128-blocks), then the solve
                // of the rows below, column tile by column tile from the right (a tile reads the ones to its left)
                size_t keep = 0;
                std::vector<Tri> later;
                for (size_t i = tri_next; i < tri.size(); ++i) {
                    const Launch& tl = tri[i].l;
                    const int span = tl.tri_mb * (tl.tiles128 ? 2 : 1) * 2;        // 64-blocks a pair covers
                    if (tri[i].ready == panel_idx && span <= ob) launches.push_back(tl);
                    else later.push_back(tri[i]);
                    (void)keep;
                }
                tri.erase(tri.begin() + tri_next, tri.end());
                tri.insert(tri.end(), later.begin(), later.end());
                if (rch < nb)
                    for (int jt = (pe - J) / 2 - 1; jt >= 0; --jt) {
                        Launch l;
                        l.kind = L_PSOLVE; l.J = J; l.pe = pe; l.c_lo = jt; l.r_lo = rch; l.tiles128 = 1;
                        l.fs.njobs = 0; l.fs.nblk = 0;
                        launches.push_back(l);
                    }
            }
            if (pe >= nb) break;
            // ... and its rows below (needed by the bulk rows of the next chain), then the far columns
            if (rch < nb) {
                rest.on = true; rest.given = 0;
                Launch& l = rest.l;
                l = Launch();
                l.kind = L_TRAIL; l.J = J; l.pe = pe; l.c_lo = pe; l.c_hi = ne; l.with_leaf = 0;
                l.tiles128 = t128 ? 1 : 0;
                l.r_lo = rch; l.r_hi = nb;
                l.fs.njobs = 0; l.fs.nblk = 0;
                rest.ntiles = t128 ? region_tiles(pe / 2, ne / 2, rch / 2, nb / 2) : region_tiles(pe, ne, rch, nb);
            }
            if (ne < nb) {
                far.on = true; far.given = 0;
                Launch& l = far.l;
                l = Launch();
                l.kind = L_TRAIL; l.J = J; l.pe = pe; l.c_lo = ne; l.c_hi = nb; l.with_leaf = 0;
                l.tiles128 = t128 ? 1 : 0;
                l.r_lo = 0; l.r_hi = nb;
                l.fs.njobs = 0; l.fs.nblk = 0;
                far.ntiles = t128 ? region_tiles(ne / 2, nb / 2, 0, nb / 2) : region_tiles(ne, nb, 0, nb);
            }
        }
        emit_tri(1L << 60, 1 << 30);          // the rest of the inverse behind the factorisation
    }

    void run() {
        if (pp.interleaved) { run_interleaved(); return; }
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
            panel(J, pe, leaf_done, next_leaf);
            if (pe >= nb) break;
            if (pe < cf) {
                Launch l;
                l.kind = L_TRAIL; l.J = J; l.pe = pe; l.c_lo = pe; l.c_hi = cf; l.tiles128 = wide128(cf) ? 1 : 0;
                l.with_leaf = next_leaf ? 1 : 0;
                l.fs.njobs = 0; l.fs.nblk = 0;
                launches.push_back(l);
                fq.end_launch();
            }
            leaf_done = next_leaf && pe < cf;
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
    PlanParams pp;
    FillQueue fq;
    int urgent_row = 0;           // CUPD rows below this 128-row block feed the next row of the inverse
    int last_dupd = -1, last_cupd = -1;
    int kb_end = 0;               // end of the last panel whose inverse jobs are queued

    bool wide_ok() const { return pp.fill_wide && (pp.ob & 1) == 0 && (pp.nb & 1) == 0; }

    // the far columns [cf, nb) of the trailing update of panel [J, pe) become the filler job with a deadline at the
    // end of the next panel's chain
    void queue_far_update(int J, int pe, int cf) {
        fq.syrk_job = -1;
        if (cf >= pp.nb) return;
        QJob jb;
        jb.j.type = FILL_SYRK;
        jb.j.R0 = 0; jb.j.R1 = pp.nb / 2; jb.j.j0 = cf; jb.j.j1 = pp.nb; jb.j.kb0 = J; jb.j.kb1 = pe;
        jb.j.wide = wide_ok() && (cf & 1) == 0 ? 1 : 0;
        jb.total = jb.j.wide ? syrk_tiles_wide(pp.nb / 2, cf, pp.nb) : syrk_tiles(pp.nb / 2, cf, pp.nb);
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
            jb.j.wide = wide_ok() && (pe & 1) == 0 && (J & 1) == 0 ? 1 : 0;
            jb.ncols = jb.j.wide ? pe / 2 : pe;
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

// ---------------------------------------------------------------------------------------------------
// The same plan as a task graph for ONE persistent launch (lcgp_hip.hip: dag_kernel).
//
// A SEGMENT is what a launch (or one filler job of a launch) is in the launch-by-launch executor: a kind, its integer
// parameters and a number of workgroup-sized TASKS, enumerated exactly like the blocks of that launch.  All tasks of
// the graph form one global sequence (segment after segment); the workgroups of the persistent kernel take them in that
// order from one counter, so a task that has been taken is held by a RUNNING workgroup and everything it may wait for
// lies before it in the sequence: the earliest unfinished task can always run, whatever number of workgroups is
// resident (no co-residency assumption, no deadlock).
// Dependencies are per (segment, component): a task of component k starts when the counters cnt[dep, k] of the
// segments listed in `dep` have reached `need` (all tasks of that segment and component have finished); when it ends it
// adds one to cnt[own segment, k].  The lists are DERIVED here from the blocks each segment reads and writes
// (rectangles of 64x64 blocks per matrix; a conflict is an overlap with at least one writer: read-after-write,
// write-after-write and write-after-read alike) and reduced transitively.  tests/test_fill_sched.py replays the graph
// on numpy matrices and checks every block-level hazard against the declared lists.
// ---------------------------------------------------------------------------------------------------
constexpr int DAG_MAXDEP = 16;

enum SegKind { S_LEAF = 1, S_STEP = 2, S_TRAIL = 3, S_FILL = 4, S_TRI = 5, S_PSOLVE = 6 };

struct DagSeg {
    int kind;
    int t0, ntasks;          // task ids [t0, t0 + ntasks)
    int per_comp;            // tasks per component (= what cnt[this segment, k] reaches)
    int k_off;               // component of task b (index within the segment): b < k_off ? b : (b - k_off) % q
    int ndeps;
    int dep[DAG_MAXDEP];     // segment indices (all smaller than this segment's)
    int need[DAG_MAXDEP];    // their per_comp
    int J, pe, c, diag_end, has_special, n_trmm, n_upd;      // S_LEAF (J) / S_STEP
    int c_lo, c_hi, tiles128, with_leaf;                     // S_TRAIL (+ J, pe)
    int t_first, t_count;                                    // S_TRAIL: a sub-range of the update's tiles (t_count = 0: all)
    int r_lo, r_hi;                                          // S_TRAIL: block rows [max(column, r_lo), r_hi) (r_hi = 0: all)
    int trmm_r0, upd_r0;                                     // S_STEP: first block row of the solve / delayed-update tiles
    int tri_mb, tri_p0, tri_np, tri_w;                       // S_TRI (+ tiles128)
    FillJob job;                                             // S_FILL
};

enum { BUF_M = 0, BUF_W = 1, BUF_V = 2, BUF_STAT = 3 };

struct Access {
    int buf, r0, r1, c0, c1;     // blocks [r0, r1) x [c0, c1) of one component's matrix
    bool write;
};

class DagBuilder {
 public:
    std::vector<DagSeg> segs;
    std::vector<std::vector<Access>> acc;     // per segment (kept for the dump / tests)
    int ntasks = 0;
    bool failed = false;

    DagBuilder(int nb_, int q_) : nb(nb_), q(q_) {}

    void build(const std::vector<Launch>& launches) {
        for (const Launch& l : launches) {
            if (l.kind == L_LEAF) {
                DagSeg s = blank(S_LEAF);
                s.J = l.J; s.pe = l.pe;
                s.ntasks = q; s.per_comp = 1; s.k_off = q;
                std::vector<Access> a;
                leaf_access(a, l.J);
                push(s, a);
            } else if (l.kind == L_STEP) {
                DagSeg s = blank(S_STEP);
                s.J = l.J; s.pe = l.pe; s.c = l.c; s.diag_end = l.diag_end; s.has_special = l.has_special;
                s.n_trmm = l.n_trmm; s.n_upd = l.n_upd;
                s.trmm_r0 = l.trmm_r0 ? l.trmm_r0 : l.c + 1;
                s.upd_r0 = l.upd_r0 ? l.upd_r0 : l.c + 2;
                s.per_comp = l.n_trmm + l.n_upd;
                s.ntasks = s.per_comp * q;
                s.k_off = l.has_special ? q : 0;
                std::vector<Access> a;
                step_access(a, l);
                if (s.ntasks > 0) push(s, a);
            } else if (l.kind == L_TRAIL) {
                DagSeg s = blank(S_TRAIL);
                s.J = l.J; s.pe = l.pe; s.c_lo = l.c_lo; s.c_hi = l.c_hi; s.tiles128 = l.tiles128; s.with_leaf = l.with_leaf;
                std::vector<Access> a;
                int nt;
                const int u = l.tiles128 ? 2 : 1, nbt = nb / u;
                const int rlo = l.r_lo / u, rhi = l.r_hi ? l.r_hi / u : nbt;
                s.r_lo = l.r_lo; s.r_hi = l.r_hi;
                if (l.t_count > 0 || l.r_lo || l.r_hi) {
                    // tiles [t_first, t_first + t_count) of the region (column-major over the tile columns from c_lo on, in
                    // column c the tile rows [max(c, r_lo), r_hi)): the blocks they touch, column by column
                    const long total = region_tiles(l.c_lo / u, l.c_hi / u, rlo, rhi);
                    const long first = l.t_count > 0 ? l.t_first : 0;
                    long left = l.t_count > 0 ? l.t_count : total;
                    nt = (int)left;
                    int c = l.c_lo / u;
                    long t = first;
                    auto col_tiles = [&](int cc) { const int lo = cc > rlo ? cc : rlo; return rhi > lo ? rhi - lo : 0; };
                    while (c < l.c_hi / u && t >= col_tiles(c)) { t -= col_tiles(c); ++c; }
                    while (left > 0 && c < l.c_hi / u) {
                        const long in_col = col_tiles(c) - t;
                        const long take = left < in_col ? left : in_col;
                        const int ra = (c > rlo ? c : rlo) + (int)t, rb = ra + (int)take;
                        a.push_back({BUF_M, ra * u, rb * u, c * u, (c + 1) * u, true});
                        a.push_back({BUF_M, ra * u, rb * u, l.J, l.pe, false});
                        a.push_back({BUF_M, c * u, (c + 1) * u, l.J, l.pe, false});
                        left -= take; t = 0; ++c;
                    }
                    if (left > 0) { failed = true; return; }
                    s.t_first = (int)first; s.t_count = nt;
                } else {
                    nt = l.tiles128 ? trapezoid_tiles(nb / 2, l.c_lo / 2, l.c_hi / 2) + (l.with_leaf ? 1 : 0)
                                    : trapezoid_tiles(nb, l.c_lo, l.c_hi);
                    a.push_back({BUF_M, l.c_lo, nb, l.J, l.pe, false});
                    a.push_back({BUF_M, l.c_lo, nb, l.c_lo, l.c_hi, true});
                    if (l.with_leaf) leaf_access(a, l.c_lo);
                }
                s.per_comp = nt;
                s.ntasks = nt * q;
                s.k_off = l.with_leaf ? q : 0;
                push(s, a);
            }
            if (l.kind == L_TRI) {
                DagSeg s = blank(S_TRI);
                s.tiles128 = l.tiles128; s.tri_mb = l.tri_mb; s.tri_p0 = l.tri_p0; s.tri_np = l.tri_np; s.tri_w = l.tri_w;
                s.per_comp = l.tri_np * l.tri_mb * l.tri_mb;
                s.ntasks = s.per_comp * q; s.k_off = 0;
                std::vector<Access> a;
                const int u = l.tiles128 ? 2 : 1, mb = l.tri_mb * u;              // in 64-blocks
                for (int pr = l.tri_p0; pr < l.tri_p0 + l.tri_np; ++pr) {
                    const int C0 = 2 * pr * mb, R0 = C0 + mb;
                    const int Re = R0 + mb < nb ? R0 + mb : nb;
                    if (R0 >= nb) continue;
                    if (l.tri_w == 0) {
                        a.push_back({BUF_M, R0, Re, C0, R0, false});
                        a.push_back({BUF_W, C0, R0, C0, R0, false});
                        a.push_back({BUF_V, R0, Re, C0, R0, true});
                    } else {
                        a.push_back({BUF_W, R0, Re, R0, Re, false});
                        a.push_back({BUF_V, R0, Re, C0, R0, false});
                        a.push_back({BUF_W, R0, Re, C0, R0, true});
                    }
                }
                push(s, a);
            }
            if (l.kind == L_PSOLVE) {
                DagSeg s = blank(S_PSOLVE);
                s.J = l.J; s.pe = l.pe; s.c_lo = l.c_lo; s.r_lo = l.r_lo; s.tiles128 = 1;
                s.per_comp = (nb - l.r_lo) / 2;
                s.ntasks = s.per_comp * q; s.k_off = 0;
                std::vector<Access> a;
                const int c0 = l.J + 2 * l.c_lo;                       // first 64-block column of the tile column
                a.push_back({BUF_M, l.r_lo, nb, l.J, c0 + 2, false});
                a.push_back({BUF_W, c0, c0 + 2, l.J, c0 + 2, false});
                a.push_back({BUF_M, l.r_lo, nb, c0, c0 + 2, true});
                if (s.ntasks > 0) push(s, a);
            }
            for (int i = 0; i < l.fs.njobs; ++i) {
                const FillJob& jb = l.fs.job[i];
                if (jb.nblk <= 0) continue;
                DagSeg s = blank(S_FILL);
                s.job = jb;
                s.ntasks = jb.nblk; s.per_comp = jb.nblk / q; s.k_off = 0;
                std::vector<Access> a;
                job_access(a, jb);
                push(s, a);
            }
        }
        derive_deps();
    }

 private:
    int nb, q;

    static DagSeg blank(int kind) {
        DagSeg s;
        s.kind = kind; s.t0 = 0; s.ntasks = 0; s.per_comp = 0; s.k_off = 0; s.ndeps = 0;
        for (int i = 0; i < DAG_MAXDEP; ++i) { s.dep[i] = -1; s.need[i] = 0; }
        s.J = s.pe = s.c = s.diag_end = s.has_special = s.n_trmm = s.n_upd = 0;
        s.c_lo = s.c_hi = s.tiles128 = s.with_leaf = 0; s.t_first = s.t_count = 0;
        s.r_lo = s.r_hi = s.trmm_r0 = s.upd_r0 = 0;
        s.tri_mb = s.tri_p0 = s.tri_np = s.tri_w = 0;
        s.job.type = FILL_NONE; s.job.nblk = 0; s.job.t0 = 0; s.job.R0 = s.job.R1 = s.job.j0 = s.job.j1 = 0;
        s.job.kb0 = s.job.kb1 = 0;
        return s;
    }

    void push(DagSeg& s, const std::vector<Access>& a) {
        s.t0 = ntasks;
        ntasks += s.ntasks;
        segs.push_back(s);
        acc.push_back(a);
    }

    // diagonal block j: factor in place, inverse into W (and the zero quadrant beside an even block), running statistics
    void leaf_access(std::vector<Access>& a, int j) const {
        a.push_back({BUF_M, j, j + 1, j, j + 1, true});
        a.push_back({BUF_W, j, j + 1, j, j + 1, true});
        if ((j & 1) == 0 && j + 1 < nb) a.push_back({BUF_W, j, j + 1, j + 1, j + 2, true});
        a.push_back({BUF_STAT, 0, 1, 0, 1, true});
    }

    void step_access(std::vector<Access>& a, const Launch& l) const {
        const int c = l.c;
        const int t0 = l.trmm_r0 ? l.trmm_r0 : c + 1, t1 = t0 + l.n_trmm;        // rows of the solve tiles
        const int u0 = l.upd_r0 ? l.upd_r0 : c + 2, u1 = u0 + l.n_upd;          // rows of the delayed-update tiles
        if (l.n_trmm > 0) {
            a.push_back({BUF_W, c, c + 1, c, c + 1, false});
            if (c > l.J) {
                a.push_back({BUF_M, c, c + 1, c - 1, c, false});                  // L[c, c-1]
                a.push_back({BUF_M, t0, t1, c - 1, c, false});                    // L[r, c-1]
            }
            a.push_back({BUF_M, t0, t1, c, c + 1, true});                        // the block column itself
            const int de = l.diag_end < t1 ? l.diag_end : t1;
            for (int r = t0; r < de; ++r) a.push_back({BUF_M, r, r + 1, r, r + 1, true});
        }
        if (l.has_special) leaf_access(a, c + 1);
        if (l.n_upd > 0) {
            a.push_back({BUF_M, u0, u1, c + 1, c + 2, true});
            if (c > l.J) {
                a.push_back({BUF_M, u0, u1, l.J, c, false});
                a.push_back({BUF_M, c + 1, c + 2, l.J, c, false});
            }
        }
    }

    // blocks touched by the tiles [t0, t0 + nblk / q) of a filler job (bounding rectangles; see the enumerations above)
    void job_access(std::vector<Access>& a, const FillJob& jb) const {
        const long n = jb.nblk / q;
        const int kb0 = jb.kb0, kb1 = jb.kb1;
        if (jb.type == FILL_SYRK) {
            // column-major over j, rows R in [j / 2, R1): per column touched one rectangle
            long t = jb.t0;
            int j = jb.j0;
            while (t >= jb.R1 - (j >> 1)) { t -= jb.R1 - (j >> 1); ++j; }
            long left = n;
            while (left > 0 && j < jb.j1) {
                const long in_col = jb.R1 - (j >> 1) - t;
                const long take = left < in_col ? left : in_col;
                const int Ra = (j >> 1) + (int)t, Rb = Ra + (int)take;
                a.push_back({BUF_M, 2 * Ra, 2 * Rb, j, j + 1, true});
                a.push_back({BUF_M, 2 * Ra, 2 * Rb, kb0, kb1, false});
                a.push_back({BUF_M, j, j + 1, kb0, kb1, false});
                left -= take; t = 0; ++j;
            }
        } else if (jb.type == FILL_BROW || jb.type == FILL_CUPD) {
            const int nc = jb.j1 - jb.j0;
            long t = jb.t0, left = n;
            while (left > 0) {
                const int R = jb.R0 + (int)(t / nc), ja = jb.j0 + (int)(t % nc);
                const long in_row = nc - (t % nc);
                const long take = left < in_row ? left : in_row;
                const int jz = ja + (int)take;
                if (jb.type == FILL_BROW) {
                    const int ke = kb1 < 2 * R + 2 ? kb1 : 2 * R + 2;
                    a.push_back({BUF_W, 2 * R, 2 * R + 2, ja, jz, true});
                    a.push_back({BUF_W, 2 * R, 2 * R + 2, kb0, ke, false});
                    a.push_back({BUF_V, kb0, ke, ja, jz, false});
                } else {
                    a.push_back({BUF_V, 2 * R, 2 * R + 2, ja, jz, true});
                    a.push_back({BUF_M, 2 * R, 2 * R + 2, kb0, kb1, false});
                    a.push_back({BUF_W, kb0, kb1, ja, jz, false});
                }
                left -= take; t += take;
            }
        } else if (jb.type == FILL_DUPD) {
            long t = jb.t0, left = n;
            while (left > 0) {
                int R = 0;
                while ((long)(R + 1) * (R + 2) <= t) ++R;
                const int ja = (int)(t - (long)R * (R + 1));
                const long in_row = 2 * R + 2 - ja;
                const long take = left < in_row ? left : in_row;
                const int jz = ja + (int)take;
                const int ks = 2 * R >= kb0 ? 2 * R : kb0;
                a.push_back({BUF_V, 2 * R, 2 * R + 2, ja, jz, true});
                a.push_back({BUF_W, ks, kb1, 2 * R, 2 * R + 2, false});
                a.push_back({BUF_W, ks, kb1, ja, jz, false});
                left -= take; t += take;
            }
        } else {
            // TRI_T / TRI_W: one level of the block inverse, pairs [j0, j0 + R1) of block size mb = R0 (the whole level:
            // a job of a level is small)
            const int mb = jb.R0;
            for (int pr = jb.j0; pr < jb.j0 + jb.R1; ++pr) {
                const int C0 = 2 * pr * mb, R0 = C0 + mb;
                const int Re = R0 + mb < nb ? R0 + mb : nb;
                if (R0 >= nb) continue;
                if (jb.type == FILL_TRI_T) {
                    a.push_back({BUF_M, R0, Re, C0, R0, false});
                    a.push_back({BUF_W, C0, R0, C0, R0, false});
                    a.push_back({BUF_V, R0, Re, C0, R0, true});
                } else {
                    a.push_back({BUF_W, R0, Re, R0, Re, false});
                    a.push_back({BUF_V, R0, Re, C0, R0, false});
                    a.push_back({BUF_W, R0, Re, C0, R0, true});
                }
            }
        }
    }

    static bool overlap(const Access& x, const Access& y) {
        return x.buf == y.buf && (x.write || y.write) && x.r0 < y.r1 && y.r0 < x.r1 && x.c0 < y.c1 && y.c0 < x.c1;
    }
    bool conflict(int s, int t) const {
        for (const Access& x : acc[s])
            for (const Access& y : acc[t])
                if (overlap(x, y)) return true;
        return false;
    }

    void derive_deps() {
        const int ns = (int)segs.size();
        const int nw = (ns + 63) / 64;
        std::vector<unsigned long long> clo((size_t)ns * nw, 0ull);      // transitive closure of the kept dependencies
        for (int s = 0; s < ns; ++s) {
            unsigned long long* cs = &clo[(size_t)s * nw];
            for (int t = s - 1; t >= 0; --t) {
                if ((cs[t >> 6] >> (t & 63)) & 1ull) continue;            // already implied
                if (!conflict(t, s)) continue;
                DagSeg& sg = segs[s];
                if (sg.ndeps >= DAG_MAXDEP) { failed = true; return; }
                sg.dep[sg.ndeps] = t;
                sg.need[sg.ndeps] = segs[t].per_comp;
                ++sg.ndeps;
                cs[t >> 6] |= 1ull << (t & 63);
                const unsigned long long* ct = &clo[(size_t)t * nw];
                for (int w = 0; w < nw; ++w) cs[w] |= ct[w];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// The ORDER of the one task sequence (dag_kernel takes the tasks in sequence order; a task that is not ready blocks the
// workgroup that took it, so the order decides how much of the chip waits).  The graph is scheduled on the host the way
// the GPU will run it: a list schedule on `slots` workgroup slots with estimated task durations, always giving a free slot
// the ready segment with the longest path to the end of the graph (the chain of diagonal blocks first, update tiles and
// the inverse as filling).  The order in which the simulation STARTS tasks is the sequence: runs (segment, first task,
// count).  A run only follows runs of everything its segment depends on, because a segment becomes ready in the
// simulation only after all tasks of its dependencies have started AND ended -- so the in-order argument of dag_kernel
// (the earliest unfinished task can always run) holds for any durations, right or wrong; wrong estimates cost waiting,
// never correctness.  Counters stay per (segment, component): the runs of a segment share them.
// ---------------------------------------------------------------------------------------------------
struct DagRun {
    int seg;       // segment
    int b0, n;     // its tasks [b0, b0 + n)
    int t0;        // position of the run's first task in the sequence
};

inline double seg_task_us(const DagSeg& s, int ob) {
    switch (s.kind) {
        case S_LEAF: return 17.0;
        case S_STEP: return s.has_special ? 22.0 : 12.0;
        case S_TRAIL: return (s.tiles128 ? 62.0 : 21.0) * (s.pe - s.J) / 4.0 + (s.with_leaf ? 10.0 : 0.0);
        case S_PSOLVE: return 35.0 * (s.c_lo + 1);
        case S_TRI: return s.tiles128 ? 10.0 + 35.0 * (s.tri_mb + 1) * 0.5 : 4.0 + 3.0 * (s.tri_mb + 1) * 0.5;
        default: return 25.0 * (s.job.kb1 - s.job.kb0 > 0 ? (s.job.kb1 - s.job.kb0) / 4.0 : 1.0);
    }
    (void)ob;
}

class DagScheduler {
 public:
    std::vector<DagRun> runs;

    // segs: in any order consistent with their dependency lists (DagBuilder's).  slots: resident workgroups.
    void run(const std::vector<DagSeg>& segs, int slots, int ob) {
        const int ns = (int)segs.size();
        std::vector<double> dur(ns), bl(ns, 0.0);
        std::vector<std::vector<int>> succ(ns);
        std::vector<int> ndep(ns, 0);
        for (int i = 0; i < ns; ++i) {
            dur[i] = seg_task_us(segs[i], ob);
            ndep[i] = segs[i].ndeps;
            for (int d = 0; d < segs[i].ndeps; ++d) succ[segs[i].dep[d]].push_back(i);
        }
        // bottom level: the longest chain of task durations from the segment to the end (a segment with more tasks than
        // slots counts its rounds)
        for (int i = ns - 1; i >= 0; --i) {
            double m = 0.0;
            for (int j : succ[i]) if (bl[j] > m) m = bl[j];
            const double rounds = (double)((segs[i].ntasks + slots - 1) / slots);
            bl[i] = m + dur[i] * (rounds > 1.0 ? rounds : 1.0);
        }
        std::vector<int> next(ns, 0), done(ns, 0);
        std::vector<int> ready;                       // indices of ready segments with tasks left
        for (int i = 0; i < ns; ++i) if (ndep[i] == 0) ready.push_back(i);
        struct Ev { double t; int seg, n; };
        std::vector<Ev> heap;                         // min-heap on t
        auto hpush = [&](Ev e) {
            heap.push_back(e);
            size_t i = heap.size() - 1;
            while (i > 0 && heap[(i - 1) / 2].t > heap[i].t) { std::swap(heap[(i - 1) / 2], heap[i]); i = (i - 1) / 2; }
        };
        auto hpop = [&]() {
            Ev top = heap[0];
            heap[0] = heap.back();
            heap.pop_back();
            size_t i = 0;
            for (;;) {
                size_t l = 2 * i + 1, r = l + 1, m = i;
                if (l < heap.size() && heap[l].t < heap[m].t) m = l;
                if (r < heap.size() && heap[r].t < heap[m].t) m = r;
                if (m == i) break;
                std::swap(heap[m], heap[i]);
                i = m;
            }
            return top;
        };
        double now = 0.0;
        int free_slots = slots, t0 = 0;
        long left = 0;
        for (int i = 0; i < ns; ++i) left += segs[i].ntasks;
        while (left > 0) {
            // hand the free slots to the ready segments, longest remaining path first
            while (free_slots > 0 && !ready.empty()) {
                int bi = 0;
                for (int i = 1; i < (int)ready.size(); ++i)
                    if (bl[ready[i]] > bl[ready[bi]] || (bl[ready[i]] == bl[ready[bi]] && ready[i] < ready[bi])) bi = i;
                const int s = ready[bi];
                int m = segs[s].ntasks - next[s];
                if (m > free_slots) m = free_slots;
                if (!runs.empty() && runs.back().seg == s && runs.back().b0 + runs.back().n == next[s]) runs.back().n += m;
                else runs.push_back({s, next[s], m, t0});
                t0 += m;
                hpush({now + dur[s], s, m});
                next[s] += m;
                free_slots -= m;
                left -= m;
                if (next[s] >= segs[s].ntasks) { ready[bi] = ready.back(); ready.pop_back(); }
            }
            if (left <= 0) break;
            if (heap.empty()) { failed = true; return; }       // (a cycle or a dangling dependency: cannot happen for a DagBuilder graph)
            // advance to the next completion (and everything that ends at the same time)
            const double t = heap[0].t;
            now = t;
            while (!heap.empty() && heap[0].t <= t) {
                const Ev e = hpop();
                free_slots += e.n;
                done[e.seg] += e.n;
                if (done[e.seg] >= segs[e.seg].ntasks)
                    for (int j : succ[e.seg])
                        if (--ndep[j] == 0) ready.push_back(j);
            }
        }
        makespan_us = now;
        while (!heap.empty()) { const Ev e = hpop(); if (e.t > makespan_us) makespan_us = e.t; }
    }
    bool failed = false;
    double makespan_us = 0.0;
};

// ---------------------------------------------------------------------------------------------------
// Hosted panels (lcgp_hip.hip: host_kernel).  Per outer panel P of `ob` = 4 block columns three launches:
//   A(P)  q chain workgroups factor the panel's whole diagonal block (and invert it), all other workgroups run DEFERRED
//         trailing updates: jobs (column panels, K range) -- 256 x 128 tiles of  M[R, c] -= sum_{k0 <= k < k1} L[R, k] L[c, k]^T;
//   B(P)  the panel solve of the rows below,  L[R, P] = X[R, P] W_PP^T  (X = the updated panel: in the scratch matrix V, where
//         C(P-1) left it; panel 0 is solved in place, one launch per block column from the right);
//   C(P)  the rank-(64 ob) update of the NEXT panel's columns only (its diagonal block back into M, the rows below into V).
// Left-looking at the outer level: column panel c receives the finished panels in groups of `defer` (one visit with
// K = 64 ob defer instead of `defer` read-modify-write passes), staggered so that every launch carries about the same
// share: column c is visited by A(P) when (c - 1 - P) is a multiple of `defer` -- which makes the visit of A(c - 1), the
// last chance before the column's own chain, a regular one.  tests/test_fill_sched.py replays the plan on numpy matrices.
// ---------------------------------------------------------------------------------------------------
constexpr int HOST_NJ = 16;

struct HostJob {
    int cp0, ncp;        // column panels cp0 .. cp0 + ncp - 1 (256 columns = two 128-column tiles each)
    int k0, k1;          // K range in 64-blocks
    int np;              // panels per side: column panel c has the 256-row blocks c .. np - 1
    int nblk;            // blocks = tiles x components (component = fastest index); tiles column panel by column panel,
                         // row block by row block, the two column tiles of a row block adjacent
};

inline long host_job_tiles(int cp0, int ncp, int np) {
    long n = 0;
    for (int c = cp0; c < cp0 + ncp; ++c) n += 2L * (np - c);
    return n;
}

struct HostPanel {
    HostPanel() { memset((void*)this, 0, sizeof(*this)); }
    int J, pe, ne;       // the panel's block columns [J, pe), the next panel's [pe, ne)  (ne = pe: the last panel)
    int njobs;
    int nhost;           // blocks of all jobs
    HostJob job[HOST_NJ];
};

class HostPlanner {
 public:
    // nb: 64-blocks per side (a multiple of ob = 4), q components, defer >= 1
    HostPlanner(int nb_, int q_, int defer_) : nb(nb_), q(q_), defer(defer_ < 1 ? 1 : defer_) {}
    std::vector<HostPanel> panels;
    bool failed = false;
    static bool applicable(int nb) { return nb >= 8 && nb % 4 == 0; }

    void run() {
        const int ob = 4, np = nb / ob;
        std::vector<int> applied(np, 0);          // panels [0, applied[c]) have been applied to column panel c
        for (int P = 0; P < np; ++P) {
            HostPanel hp;
            hp.J = P * ob; hp.pe = hp.J + ob; hp.ne = P + 1 < np ? hp.pe + ob : hp.pe;
            for (int c = P + 1; c < np; ++c) {
                if (applied[c] >= P) continue;
                if (c != P + 1 && (c - 1 - P) % defer != 0) continue;
                // merge with the previous job when it is the column panel next to it with the same K range
                if (hp.njobs > 0) {
                    HostJob& pj = hp.job[hp.njobs - 1];
                    if (pj.cp0 + pj.ncp == c && pj.k0 == applied[c] * ob) {
                        ++pj.ncp;
                        applied[c] = P;
                        continue;
                    }
                }
                if (hp.njobs >= HOST_NJ) continue;        // (stays pending: a later launch takes it with a longer K)
                HostJob& j = hp.job[hp.njobs++];
                j.cp0 = c; j.ncp = 1; j.k0 = applied[c] * ob; j.k1 = P * ob; j.np = np;
                applied[c] = P;
            }
            if (P + 1 < np && applied[P + 1] != P) { failed = true; return; }
            // longest K first
            for (int i = 1; i < hp.njobs; ++i)
                for (int j = i; j > 0 && hp.job[j].k1 - hp.job[j].k0 > hp.job[j - 1].k1 - hp.job[j - 1].k0; --j)
                    std::swap(hp.job[j], hp.job[j - 1]);
            for (int i = 0; i < hp.njobs; ++i) {
                hp.job[i].nblk = (int)(host_job_tiles(hp.job[i].cp0, hp.job[i].ncp, np) * q);
                hp.nhost += hp.job[i].nblk;
            }
            panels.push_back(hp);
            if (P + 1 < np) applied[P + 1] = P + 1;      // C(P)
        }
    }

 private:
    int nb, q, defer;
};

}  // namespace lcgp_fill

#endif
