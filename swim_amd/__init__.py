"""swim_amd -- MI355X-native SWIM tick simulator (host-side mirror of jpfuentes2/swim's
Config / Liveness / Message surface over the C ABI in include/swimsim.h)."""
from .types import (Ack, Alive, Broadcast, Config, Dead, IndirectPing, Liveness, Member, MembershipEvent, Ping, SimConfig,
                    Suspect, isAlive, isDead, memberId, memberName, milliseconds, notAlive,
                    parseConfig, removeDeadNodes)
from .sim import Sim, SwimError


def configure(sim_config=None):
    """configure :: IO (Either Error Store)  (src/Util.hs:103-107) for the whole population
    on the GPU.  Returns (error, None) or (None, Sim)."""
    from . import _lib
    return Sim.configure(_lib.load(), sim_config or SimConfig())


def simulate(sim_config=None):
    """Like configure but raises SwimError on failure."""
    from . import _lib
    return Sim.create(_lib.load(), sim_config or SimConfig())
