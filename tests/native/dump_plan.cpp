// Prints the launch plan of one factorisation (lcgp_amd/csrc/fill_sched.h: Planner, HostPlanner) as text, for the CPU
// replay in tests/test_fill_sched.py.  Host-only: g++ -std=c++17 -I lcgp_amd/csrc tests/native/dump_plan.cpp
//   usage: dump_plan nb q ob syrk_small_tiles fill_leaf fill_step leaf_in_wide progressive [far_rides [with_dupd [host_from [defer]]]]
// One line per launch ("L key=value ..."), followed by its filler jobs ("J ...").  host_from >= 0: the launch-by-launch plan
// stops in front of that block column and the hosted panels follow, one line per panel ("H ...") followed by its
// deferred-update jobs ("U ...").
#include <cstdio>
#include <cstdlib>

#include "fill_sched.h"

static void print_job(const lcgp_fill::FillJob& j) {
    printf(" type=%d nblk=%d jt0=%d R0=%d R1=%d j0=%d j1=%d kb0=%d kb1=%d", j.type, j.nblk, j.t0, j.R0, j.R1, j.j0, j.j1, j.kb0, j.kb1);
}

int main(int argc, char** argv) {
    if (argc < 9 || argc > 13) { fprintf(stderr, "usage: dump_plan nb q ob syrk_small fill_leaf fill_step leaf_in_wide progressive\n"); return 2; }
    lcgp_fill::PlanParams pp;
    pp.nb = atoi(argv[1]); pp.q = atoi(argv[2]); pp.ob = atoi(argv[3]); pp.syrk_small_tiles = atoi(argv[4]);
    pp.fill_leaf = atoi(argv[5]); pp.fill_step = atoi(argv[6]); pp.leaf_in_wide = atoi(argv[7]);
    pp.progressive = atoi(argv[8]) != 0;
    pp.far_rides = argc > 9 ? atoi(argv[9]) != 0 : true;
    pp.with_dupd = argc > 10 ? atoi(argv[10]) != 0 : true;
    const int host_from = argc > 11 ? atoi(argv[11]) : -1;
    const int defer = argc > 12 ? atoi(argv[12]) : 2;
    if (host_from >= 0) { pp.stop_block = host_from; pp.progressive = false; }
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
    if (host_from >= 0) {
        if (!lcgp_fill::HostPlanner::applicable(pp.nb) || host_from % 4) { printf("FAILED\n"); return 1; }
        lcgp_fill::HostPlanner hp(pp.nb, pp.q, defer, host_from / 4);
        hp.run();
        if (hp.failed) { printf("FAILED\n"); return 1; }
        for (const lcgp_fill::HostPanel& p : hp.panels) {
            printf("H J=%d pe=%d ne=%d njobs=%d nhost=%d\n", p.J, p.pe, p.ne, p.njobs, p.nhost);
            for (int i = 0; i < p.njobs; ++i)
                printf("U cp0=%d ncp=%d k0=%d k1=%d np=%d nblk=%d\n", p.job[i].cp0, p.job[i].ncp, p.job[i].k0, p.job[i].k1, p.job[i].np,
                       p.job[i].nblk);
        }
    }
    return 0;
}
