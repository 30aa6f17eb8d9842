// Prints the launch plan of one factorisation (lcgp_amd/csrc/fill_sched.h: Planner) as text, for the CPU replay in
// tests/test_fill_sched.py.  Host-only: g++ -std=c++17 -I lcgp_amd/csrc tests/native/dump_plan.cpp
//   usage: dump_plan nb q ob syrk_small_tiles fill_leaf fill_step leaf_in_wide progressive [far_rides [with_dupd [pair_tiles]]]
// One line per launch ("L key=value ..."), followed by its filler jobs ("J ...").
#include <cstdio>
#include <cstdlib>

#include "fill_sched.h"

static void print_job(const lcgp_fill::FillJob& j) {
    printf(" type=%d nblk=%d jt0=%d R0=%d R1=%d j0=%d j1=%d kb0=%d kb1=%d", j.type, j.nblk, j.t0, j.R0, j.R1, j.j0, j.j1, j.kb0, j.kb1);
}

int main(int argc, char** argv) {
    if (argc < 9 || argc > 12) { fprintf(stderr, "usage: dump_plan nb q ob syrk_small fill_leaf fill_step leaf_in_wide progressive\n"); return 2; }
    lcgp_fill::PlanParams pp;
    pp.nb = atoi(argv[1]); pp.q = atoi(argv[2]); pp.ob = atoi(argv[3]); pp.syrk_small_tiles = atoi(argv[4]);
    pp.fill_leaf = atoi(argv[5]); pp.fill_step = atoi(argv[6]); pp.leaf_in_wide = atoi(argv[7]);
    pp.progressive = atoi(argv[8]) != 0;
    pp.far_rides = argc > 9 ? atoi(argv[9]) != 0 : true;
    pp.with_dupd = argc > 10 ? atoi(argv[10]) != 0 : true;
    pp.pair_tiles = argc > 11 ? atoi(argv[11]) : 0;
    lcgp_fill::Planner plan(pp);
    plan.run();
    if (plan.failed) { printf("FAILED\n"); return 1; }
    for (const lcgp_fill::Launch& l : plan.launches) {
        printf("L kind=%d J=%d pe=%d c=%d diag_end=%d has_special=%d n_trmm=%d n_upd=%d c_lo=%d c_hi=%d tiles128=%d "
               "with_leaf=%d njobs=%d nblk=%d\n", l.kind, l.J, l.pe, l.c, l.diag_end, l.has_special, l.n_trmm, l.n_upd, l.c_lo, l.c_hi,
               l.tiles128, l.with_leaf, l.fs.njobs, l.fs.nblk);
        for (int i = 0; i < l.fs.njobs; ++i) {
            printf("J");
            print_job(l.fs.job[i]);
            printf("\n");
        }
    }
    return 0;
}
