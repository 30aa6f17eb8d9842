// Prints the launch plan of one factorisation (lcgp_amd/csrc/fill_sched.h: Planner) as text, for the CPU replay in
// tests/test_fill_sched.py.  Host-only: g++ -std=c++17 -I lcgp_amd/csrc tests/native/dump_plan.cpp
//   usage: dump_plan nb q ob syrk_small_tiles fill_leaf fill_step leaf_in_wide progressive [far_rides [with_dupd [dag [interleaved]]]]
// With dag != 0 the task graph of the same plan (DagBuilder) follows the launch list:
//   S kind t0 ntasks per_comp k_off J pe c diag_end has_special n_trmm n_upd c_lo c_hi tiles128 with_leaf
//     job.type job.nblk job.t0 job.R0 job.R1 job.j0 job.j1 job.kb0 job.kb1 t_first t_count ndeps dep...
#include <cstdio>
#include <cstdlib>

#include "fill_sched.h"

int main(int argc, char** argv) {
    if (argc < 9 || argc > 13) { fprintf(stderr, "usage: dump_plan nb q ob syrk_small fill_leaf fill_step leaf_in_wide progressive\n"); return 2; }
    lcgp_fill::PlanParams pp;
    pp.nb = atoi(argv[1]); pp.q = atoi(argv[2]); pp.ob = atoi(argv[3]); pp.syrk_small_tiles = atoi(argv[4]);
    pp.fill_leaf = atoi(argv[5]); pp.fill_step = atoi(argv[6]); pp.leaf_in_wide = atoi(argv[7]);
    pp.progressive = atoi(argv[8]) != 0;
    pp.far_rides = argc > 9 ? atoi(argv[9]) != 0 : true;
    pp.with_dupd = argc > 10 ? atoi(argv[10]) != 0 : true;
    const bool dag = argc > 11 && atoi(argv[11]) != 0;
    pp.interleaved = argc > 12 && atoi(argv[12]) != 0;
    lcgp_fill::Planner plan(pp);
    plan.run();
    if (plan.failed) { printf("FAILED\n"); return 1; }
    for (const lcgp_fill::Launch& l : plan.launches) {
        printf("L %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", l.kind, l.J, l.pe, l.c, l.diag_end, l.has_special, l.n_trmm,
               l.n_upd, l.c_lo, l.c_hi, l.tiles128, l.with_leaf, l.fs.njobs, l.fs.nblk, l.t_first, l.t_count);
        for (int i = 0; i < l.fs.njobs; ++i) {
            const lcgp_fill::FillJob& j = l.fs.job[i];
            printf("J %d %d %d %d %d %d %d %d %d\n", j.type, j.nblk, j.t0, j.R0, j.R1, j.j0, j.j1, j.kb0, j.kb1);
        }
    }
    if (dag) {
        lcgp_fill::DagBuilder db(pp.nb, pp.q);
        db.build(plan.launches);
        if (db.failed) { printf("FAILED\n"); return 1; }
        for (const lcgp_fill::DagSeg& s : db.segs) {
            printf("S %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", s.kind, s.t0, s.ntasks,
                   s.per_comp, s.k_off, s.J, s.pe, s.c, s.diag_end, s.has_special, s.n_trmm, s.n_upd, s.c_lo, s.c_hi, s.tiles128,
                   s.with_leaf, s.job.type, s.job.nblk, s.job.t0, s.job.R0, s.job.R1, s.job.j0, s.job.j1, s.job.kb0, s.job.kb1,
                   s.t_first, s.t_count, s.ndeps);
            for (int i = 0; i < s.ndeps; ++i) printf(" %d:%d", s.dep[i], s.need[i]);
            printf("\n");
        }
    }
    return 0;
}
