"""BASELINE config 5 on one GPU: 30 % message loss x churn {0, 0.1, 1} % of the members crash-and-rejoin per 100 ticks,
reporting what SURVEY.md 8(d) asks of it -- the dissemination ticks-to-all of real crashes and the false-positive Dead
count -- next to ms/tick.  One JSON line per case.

  ticks-to-all: 8 members that are not part of the churn crash at tick T0 and stay down; swimsim_coverage(subject, Dead, 0)
      is polled after every tick: the first tick at which every up member holds Dead@0 (or something stronger) about the
      subject, minus T0.  null = not within the run.
  false-positive Dead: the counter false_deads (suspicion timers that fired about a member that was up, per observer --
      whatever the incarnation: under churn this includes observers burying the old incarnation of a member that is back).

usage (GPU box): config5.py [members ...]      default: 16384 32768  (a dense view stops there: 65 536 members at 30 % loss
need more view rows than a handle has).  CAP=<C>: bounded member maps (view_cap = C; include/swimsim.h) -- what runs config 5 at
its per-GPU size: `CAP=64 config5.py 2097152`.   ORACLE=1: the same run on the CPU oracle (all host threads), numbers
compared;  LOSS=<ppm>, TICKS=<n>, T0=<tick>, ROWS=<max subjects>, CHURN=<per-mille list>, S=<suspicion ticks> override the defaults.
(S: with bounded maps an entry lives ~C / (changes per member-tick) ticks; a suspicion timeout beyond that never fires -- DESIGN.md
section 6 -- so the capacity sweep is run with timeouts on both sides of the entries' lifetime.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, _abi, _lib, workloads

LOSS = int(os.environ.get("LOSS", 300000))
T = int(os.environ.get("TICKS", 300))
T0 = int(os.environ.get("T0", 100))
NTRACK = 8


def run(abi, n, per_mille, threads=0):
    cap = int(os.environ.get("CAP", 0))
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=1, lossPpm=LOSS, maxSubjects=int(os.environ.get("ROWS", min(n, 60000))), eventMask=0x10 if cap else 0,
                   gcTicks=_abi.GC_AUTO if (per_mille and not cap) else 0, viewCap=cap, suspicionTicks=int(os.environ.get("S", 0)))
    s = Sim.create(abi, sc)
    if threads:                                      # the oracle only (scales to ~32 threads)
        from tests import oracle_binding
        oracle_binding.set_threads(s, min(32, threads))
    churn = workloads.hashed_crashes(n, 9, per_mille * T // 100, 1000, 5, T - 60) if per_mille else []
    busy = {m for (_, m) in churn}
    for (t, m) in churn:
        s.crash(m, t); s.scheduleFault(t + 50, m, True)
    tracked = [m for m in range(n // 7, n) if m not in busy][:NTRACK]
    for m in tracked:
        s.crash(m, T0)
    t0 = time.time(); s.step(T0); warm = time.time() - t0
    c_warm = s.counters()
    to_all = {m: None for m in tracked}
    cover = []
    for _ in range(T - T0):
        s.step(1)
        row = []
        for m in tracked:
            hold, up = s.coverage(m, 2, 0)
            row.append(hold)
            if to_all[m] is None and hold == up:
                to_all[m] = s.tick - T0
        cover.append(row)
    c = s.counters()
    hold_end, up_end = s.coverage(tracked[0], 2, 0)
    out = {"members": n, "loss_ppm": LOSS, "view_cap": cap, "churn_percent_per_100_ticks": per_mille / 10.0, "ticks": T, "crash_rejoin_pairs": len(churn),
           "evicted_per_member_tick": round(c.get("evicted", 0) / float(n * T), 3), "changes_per_member_tick": round(c["changes"] / float(n * T), 2),
           "payloads_per_member_tick": round(c["payloads"] / float(n * T), 2),
           "ms_per_tick_first_%d_ticks" % T0: round(warm / T0 * 1e3, 3),
           "false_positive_dead": c["false_deads"], "false_positive_dead_per_member_tick": round(c["false_deads"] / float(n * T), 4),
           "false_suspects": c["false_suspects"], "timers_fired": c["timers_fired"], "refutes": c["refutes"],
           "tracked_crashes": len(tracked), "dissemination_ticks_to_all": [to_all[m] for m in tracked],
           "coverage_at_end_of_first_tracked": [hold_end, up_end],
           "suspicion_ticks": s.resolved.suspicion_ticks, "digest": "%016x" % s.digest()}
    s.close()
    return out, cover


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768]
    abi = _lib.load()
    for n in sizes:
        for pm in [int(x) for x in os.environ.get("CHURN", "0,1,10").split(",")]:   # per mille of the members per 100 ticks
            out, cover = run(abi, n, pm)
            if os.environ.get("ORACLE"):
                from tests import oracle_binding
                ref, rcover = run(oracle_binding.load(), n, pm, threads=os.cpu_count())
                keys = ("false_positive_dead", "false_suspects", "timers_fired", "refutes", "dissemination_ticks_to_all", "digest")
                out["verified_vs_oracle"] = all(out[k] == ref[k] for k in keys) and cover == rcover
                out["oracle_ms_per_tick_first_%d_ticks" % T0] = ref["ms_per_tick_first_%d_ticks" % T0]
                out["oracle_threads"] = min(32, os.cpu_count())
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
