// Prints the launch plan of one factorisation (lcgp_amd/csrc/fill_sched.h: Planner) as text, for the CPU replay in
// tests/test_fill_sched.py.  Host-only: g++ -std=c++17 -I lcgp_amd/csrc tests/native/dump_plan.cpp
//   usage: dump_plan nb q ob syrk_small_tiles fill_leaf fill_step leaf_in_wide progressive [far_rides [with_dupd [dag [interleaved [with_trtri [trtri_all_small [fill_wide]]]]]]]
// One line per launch ("L key=value ..."), followed by its filler jobs ("J ..."); with dag != 0 the task graph of the same
// plan (DagBuilder) follows, one line per segment ("S key=value ... deps=seg:need,seg:need").
#include <cstdio>
#include <cstdlib>

#include "fill_sched.h"

static void print_job(const lcgp_fill::FillJob& j) {
    printf(" type=%d nblk=%d jt0=%d R0=%d R1=%d j0=%d j1=%d kb0=%d kb1=%d wide=%d", j.type, j.nblk, j.t0, j.R0, j.R1, j.j0, j.j1, j.kb0, j.kb1, j.wide);
}

int main(int argc, char** argv) {
    if (argc < 9 || argc > 16) { fprintf(stderr, "usage: dump_plan nb q ob syrk_small fill_leaf fill_step leaf_in_wide progressive\n"); return 2; }
    lcgp_fill::PlanParams pp;
    pp.nb = atoi(argv[1]); pp.q = atoi(argv[2]); pp.ob = atoi(argv[3]); pp.syrk_small_tiles = atoi(argv[4]);
    pp.fill_leaf = atoi(argv[5]); pp.fill_step = atoi(argv[6]); pp.leaf_in_wide = atoi(argv[7]);
    pp.progressive = atoi(argv[8]) != 0;
    pp.far_rides = argc > 9 ? atoi(argv[9]) != 0 : true;
    pp.with_dupd = argc > 10 ? atoi(argv[10]) != 0 : true;
    const bool dag = argc > 11 && atoi(argv[11]) != 0;
    pp.interleaved = argc > 12 && atoi(argv[12]) != 0;
    pp.with_trtri = argc > 13 && atoi(argv[13]) != 0;
    pp.trtri_all_small = argc > 14 ? atoi(argv[14]) : 0;
    pp.fill_wide = argc > 15 && atoi(argv[15]) != 0;
    lcgp_fill::Planner plan(pp);
    plan.run();
    if (plan.failed) { printf("FAILED\n"); return 1; }
    for (const lcgp_fill::Launch& l : plan.launches) {
        printf("L kind=%d J=%d pe=%d c=%d diag_end=%d has_special=%d n_trmm=%d n_upd=%d trmm_r0=%d upd_r0=%d c_lo=%d c_hi=%d tiles128=%d "
               "with_leaf=%d t_first=%d t_count=%d r_lo=%d r_hi=%d tri_mb=%d tri_p0=%d tri_np=%d tri_w=%d njobs=%d nblk=%d\n", l.kind, l.J,
               l.pe, l.c, l.diag_end, l.has_special, l.n_trmm, l.n_upd, l.trmm_r0, l.upd_r0, l.c_lo, l.c_hi, l.tiles128, l.with_leaf,
               l.t_first, l.t_count, l.r_lo, l.r_hi, l.tri_mb, l.tri_p0, l.tri_np, l.tri_w, l.fs.njobs, l.fs.nblk);
        for (int i = 0; i < l.fs.njobs; ++i) {
            printf("J");
            print_job(l.fs.job[i]);
            printf("\n");
        }
    }
    if (dag) {
        lcgp_fill::DagBuilder db(pp.nb, pp.q);
        db.build(plan.launches);
        if (db.failed) { printf("FAILED\n"); return 1; }
        for (const lcgp_fill::DagSeg& s : db.segs) {
            printf("S kind=%d t0=%d ntasks=%d per_comp=%d k_off=%d J=%d pe=%d c=%d diag_end=%d has_special=%d n_trmm=%d n_upd=%d trmm_r0=%d "
                   "upd_r0=%d c_lo=%d c_hi=%d tiles128=%d with_leaf=%d t_first=%d t_count=%d r_lo=%d r_hi=%d tri_mb=%d tri_p0=%d tri_np=%d tri_w=%d",
                   s.kind, s.t0, s.ntasks, s.per_comp, s.k_off, s.J, s.pe, s.c, s.diag_end, s.has_special, s.n_trmm, s.n_upd, s.trmm_r0,
                   s.upd_r0, s.c_lo, s.c_hi, s.tiles128, s.with_leaf, s.t_first, s.t_count, s.r_lo, s.r_hi, s.tri_mb, s.tri_p0, s.tri_np,
                   s.tri_w);
            print_job(s.job);
            printf(" ndeps=%d deps=", s.ndeps);
            for (int i = 0; i < s.ndeps; ++i) printf("%s%d:%d", i ? "," : "", s.dep[i], s.need[i]);
            printf("\n");
        }
        // the order of the one task sequence: a list schedule of the graph (DagScheduler); runs of (segment, first task, count)
        lcgp_fill::DagScheduler sch;
        sch.run(db.segs, 512, pp.ob);
        if (sch.failed) { printf("FAILED\n"); return 1; }
        for (const lcgp_fill::DagRun& r : sch.runs) printf("R seg=%d b0=%d n=%d t0=%d\n", r.seg, r.b0, r.n, r.t0);
        printf("M makespan_us=%d\n", (int)sch.makespan_us);
    }
    return 0;
}
