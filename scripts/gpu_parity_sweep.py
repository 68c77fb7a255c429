"""Randomised parity sweep on the real GPU: libswimsim.so against the oracle over random sizes (2 .. 65 536), probe
counts, loss, fault schedules with rejoins, both target schemes and 1-8 shards (LocalFabric); every observable every
10 ticks, for SECONDS (default 150) s; every option of the handle: settling, join pull, periodic state pull, rumours injected
from outside, the three ways explicit records are processed.  Oracle on up to 32 threads.
usage (GPU box): [SECONDS=480] [SEED=12345] python scripts/gpu_parity_sweep.py"""
import sys, random, time
import os; sys.path.insert(0, os.getcwd())
from swim_amd import Sim, SimConfig, Config, workloads, _abi
from swim_amd.shard import LocalFabric, ShardedSim
from tests import hostemu_binding, oracle_binding
from swim_amd import _lib
emu=_lib.load(); orc=oracle_binding.load()
rng=random.Random(int(os.environ.get('SEED', 12345)))
BUDGET=int(os.environ.get('SECONDS', 150))
t0=time.time(); runs=0
while time.time()-t0 < BUDGET:
    n=rng.choice([2,3,17,64,129,300,777,1024,4096,10000,65536])
    p=rng.choice([1,2,3,3,3,5,10])
    loss=rng.choice([0,0,0,10000,100000,300000]) if n <= 4096 else rng.choice([0,0,2000])
    scheme=rng.choice([0,0,1])
    shards=1
    if scheme==0 and n>=64 and rng.random()<0.4:
        shards=rng.choice([g for g in (2,3,4,8) if n%g==0] or [1])
    seed=rng.randrange(1,1<<30)
    sc=SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F if n<=4096 else 0, suspicionTicks=rng.choice([3,6,12]),
                 maxSubjects=min(n,4096), targetScheme=scheme, inboxCap=rng.choice([0,0,1,2]) if n <= 4096 else 0,
                 gcTicks=_abi.GC_AUTO if rng.random()<0.5 else 0, joinPull=1 if rng.random()<0.5 else 0,
                 pullTicks=rng.choice([0,0,2,5,17,60]) if shards==1 else rng.choice([0,0,0,3,17]))
    sc.pushPull = bool(sc.pullTicks) and rng.random() < 0.5                      # (on shards too since round 6)
    if rng.random() < 0.25:
        sc.strictReferenceRules = True                                          # the literal rule: on shards, with settling and state pulls too since round 6
    rk=rng.choice(["","0","1"])
    if rk: os.environ["SWIMSIM_RECORDS_KERNEL"]=rk
    else: os.environ.pop("SWIMSIM_RECORDS_KERNEL", None)
    inject = rng.random()<0.4                                                   # (sharded clusters: ShardedSim.injectRumor routes and notes)
    print("cfg", n,p,loss,scheme,shards,seed, "gc", sc.gcTicks!=0, "jp", sc.joinPull, "pull", sc.pullTicks, "rk"+rk, "inject", inject, "push", sc.pushPull, "strict", sc.strictReferenceRules, flush=True)
    a=Sim.create(orc, sc)
    oracle_binding.set_threads(a, min(32, os.cpu_count() or 1))
    b=Sim.create(emu, sc) if shards==1 else ShardedSim(emu, sc, LocalFabric(shards), device="cuda:0")
    faults=[]
    for k in range(rng.randrange(0, min(256, max(1,n//8))+1)):
        m=rng.randrange(n); t=rng.randrange(1,40)
        faults.append((t,m,False))
        if rng.random()<0.5: faults.append((t+rng.randrange(1,30),m,True))
    for s in (a,b):
        for (t,m,up) in faults: s.scheduleFault(t,m,up)
    ticks=rng.choice([30,60,90])
    for _ in range(ticks//10):
        if inject:
            for _j in range(rng.randrange(0, 12)):
                o_, s_, st_, inc_ = rng.randrange(n), rng.randrange(n), rng.randrange(3), rng.randrange(3)
                a.injectRumor(o_, s_, st_, inc_); b.injectRumor(o_, s_, st_, inc_)
        a.step(10); b.step(10)
        ca, cb = a.counters(), b.counters()
        da, db = ca.pop("events_dropped"), cb.pop("events_dropped")
        if da or db: a.drainEventsRaw(); b.drainEventsRaw()   # a full ring: what is lost is implementation-defined
        assert ca==cb, ("counters", n,p,loss,scheme,shards,seed)
        assert a.digest()==b.digest(), ("digest", n,p,loss,scheme,shards,seed)
        assert a.drainEventsRaw()==b.drainEventsRaw(), ("events", n,p,loss,scheme,shards,seed)
    assert a.firstDetection()==b.firstDetection()
    a.close(); b.close(); runs+=1
print("randomized parity sweep ok:", runs, "configurations")
