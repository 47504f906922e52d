"""
MI355X-native batched NLP solver for the receding-horizon optimisation of
TGoldC/Motion-Planning-for-Autonomous-Driving-with-MPC (MPC_Planner/optimizer.py).

  solver.BatchedMPCSolver ....... owner of the C-ABI handle (include/mpcgpu.h -> csrc/libmpcgpu.so, HIP/gfx950)
  optimizer.CasadiOptimizer ..... look-alike of the reference class (constructor, solver(), optimize())
  optimizer.ForcesproOptimizer .. call-surface twin (solver.solve(problem))
  mpc_planner.MPCPlanner ........ the caller (mpc_planner.py:21-75, 296-314) without plotting: plan(), result files, collision check;
                                  its metrics on the device: deviation / RMSD / circle clearance (mpc_planner.py:184-199, 279-292)
  noise ......................... seeded stand-in for the reference's `noised: True` draws (mirror of the device generator)
  scenario ...................... CommonRoad XML + settings -> planning configuration (configuration.py:400-623) on numpy

The directory name contains hyphens (it is fixed by the build contract); import it with
`importlib.import_module("motion-planning-for-autonomous-driving-with-mpc_amd")` or through the `mpc_amd` shim
at the repository root.
"""
from ._abi import MpcLibraryError, load_library  # noqa: F401
from .solver import BatchedMPCSolver, MpcError, SolveResult, rescue_failed  # noqa: F401
from . import mpc_planner, noise, optimizer, scenario, sharding  # noqa: E402,F401
from .optimizer import CasadiOptimizer, ForcesproOptimizer  # noqa: E402,F401
from .mpc_planner import MPCPlanner  # noqa: E402,F401
