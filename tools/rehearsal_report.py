#!/usr/bin/env python3
"""tools/rehearsal_report.py <bench line .json> [...] — one page for whoever gets the first multi-GPU lease: every rank's observed values of a `bench.py --gpus N`
line next to what DESIGN.md §5.5 expects at `-s 80` per rank, with a PASS / LOOK verdict per line (LOOK = outside the expectation: the first thing to chase; on a
box with fewer GPUs than ranks the transport lines say LOOK by construction — the halos are staged through host memory there)."""
import json
import sys

# per rank at -s 80 per rank: grid -> (halo bytes per step, ghosts)
GRID = {1: (0.0, 277000), 2: (3.15e6, 277000), 4: (6.0e6, 277000), 8: (9.0e6, 277000)}
FORCE_MS, NEIGH_MS, WINDOW_COMM_MS = 0.170, 0.60, 0.27


def verdict(ok):
    return "PASS" if ok else "LOOK"


def report(path, out=sys.stdout):
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        print("%s: no JSON line (bench.py failed: see its stderr)" % path, file=out)
        return False
    d = json.loads(lines[-1])
    n, steps = d["n_gpus"], d["steps"]
    size_ok = "-s 80 per GPU" in d["config"]["workload"]
    halo_exp, ghosts_exp = GRID.get(n, (None, 277000))
    nreb = max(steps // 20, 1)
    allok = True

    def line(what, seen, expect, ok):
        nonlocal allok
        allok = allok and ok
        print("  %-44s %-34s expected %-40s %s" % (what, seen, expect, verdict(ok)), file=out)

    print("== %s: N = %d, %d steps, %.1f Matom-steps/s, %.4f ms/step" % (path, n, steps, d["value"], d["ms_per_step"]), file=out)
    line("halos over RCCL (valid)", str(d["valid"]), "true", d["valid"] is True)
    line("transport / ranks", "%s / %d" % (d["config"]["transport"], d["config"]["transport_ranks"]), ("rccl / %d" % n) if n > 1 else "none / 1",
         (d["config"]["transport"] == ("rccl" if n > 1 else "none")) and d["config"]["transport_ranks"] == n)
    if d.get("reason"):
        print("  reason: %s" % d["reason"], file=out)
    line("host syncs per re-neighboring (max rank)", "%.1f" % d["host_syncs_per_rebuild"], "2 (3 in the window of the overlap trial)", d["host_syncs_per_rebuild"] <= 3)
    if n > 1:
        line("weak-scaling step time", "%.4f ms" % d["ms_per_step"], "<= 0.235 ms (efficiency >= 0.90 of 0.2117)", d["ms_per_step"] <= 0.235)
    ranks = d.get("per_rank") or []
    if not size_ok:
        print("  (not -s 80 per rank: the per-rank expectations below are those of -s 80)", file=out)
    for r in ranks:
        c, ph = r.get("counters", {}), r["phases_s"]
        print(" rank %d" % r.get("rank", -1), file=out)
        line("owned atoms", "%d" % r["nlocal"], "2 048 000 +- 0.1 %", abs(r["nlocal"] - 2048000) <= 2048)
        line("ghost atoms", "%d" % r["nghost"], "~%d +- 4 %%" % ghosts_exp, abs(r["nghost"] - ghosts_exp) <= 0.04 * ghosts_exp)
        if halo_exp is not None and n > 1:
            hb = r["bytes_sent"] / max(steps, 1)
            line("halo bytes per step", "%.2f MB" % (hb / 1e6), "%.1f MB +- 5 %%" % (halo_exp / 1e6), abs(hb - halo_exp) <= 0.05 * halo_exp)
        line("host syncs per re-neighboring", "%.1f" % (r["host_syncs"] / nreb), "<= 3", r["host_syncs"] / nreb <= 3)
        line("debug-transport waits per step", "%.1f" % (r["transport_syncs"] / max(steps, 1)), "0 (RCCL has none)", r["transport_syncs"] == 0)
        line("Force::compute per step", "%.4f ms" % (ph["force"] / steps * 1e3), "%.3f ms +- 15 %%" % FORCE_MS, abs(ph["force"] / steps * 1e3 - FORCE_MS) <= 0.15 * FORCE_MS)
        line("Neighbor::build per re-neighboring", "%.3f ms" % (ph["neigh"] / nreb * 1e3), "~%.2f ms (<= 0.8)" % NEIGH_MS, ph["neigh"] / nreb * 1e3 <= 0.8)
        if n > 1:
            halo_ms = (ph["comm"] - nreb * WINDOW_COMM_MS * 1e-3) / max(steps - nreb, 1) * 1e3
            line("step halo (comm minus the windows' share)", "%.4f ms" % halo_ms, "pack 7 us + transfer 25-40 us: <= 0.06 ms", halo_ms <= 0.06)
        line("buckets partition the wall clock", "%.4f + %.4f + %.4f <= %.4f s" % (ph["force"], ph["neigh"], ph["comm"], ph["total"]), "t_other >= 0 (ref/ljs.cpp:485-495)",
             ph["force"] + ph["neigh"] + ph["comm"] <= ph["total"] * (1 + 1e-9) + 1e-9)
        if c:
            line("exchange: fixed-size messages overflowed", "%d" % c["exchange_overflows"], "0", c["exchange_overflows"] == 0)
            if n > 1:
                line("borders as one exchange / swap by swap", "%d / %d" % (c["borders_direct"], c["borders_general"]), ">= 1 / <= 1 (the set-up's)", c["borders_direct"] >= 1 and c["borders_general"] <= 1)
                line("RCCL bring-up check: partners, time", "%d, %.1f ms" % (c["rccl_check_partners"], c["rccl_check_us"] / 1e3), "%d partners" % min(n - 1, 26),
                     c["rccl_check_partners"] == min(n - 1, 26))
                print("  overlap trial: without %.4f ms, with %.4f ms per step (summed over ranks) -> %s" % (
                    c["overlap_trial_off_ns"] * 1e-6, c["overlap_trial_on_ns"] * 1e-6, {1: "overlap", 0: "no overlap", -1: "undecided (window too short for the trial)"}[c["overlap_choice"]]), file=out)
    print("  => %s" % ("every line PASS" if allok else "lines marked LOOK are the first things to chase (DESIGN.md §5.5)"), file=out)
    return allok


if __name__ == "__main__":
    for p in sys.argv[1:]:
        report(p)
