"""molly.jl_amd — MI355X-native nonbonded force engine behind Molly.jl's API (imported as `molly_jl_amd`).

Holds only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/mollyhip.h, built into
libmollyhip.so) and the host-side mirror of the reference interface (`api.py`, `domain.py`).
"""
from ._lib import MollyHipError, build, device_count, lib, LIB_PATH, SIGNATURES, Stats, Config, Interactions  # noqa: F401
from .api import (  # noqa: F401
    AndersenThermostat, Atom, BOLTZMANN, COULOMB_CONST, CellListMapNeighborFinder, Coulomb, CoulombEwald, CoulombReactionField,
    CubicBoundary, CubicSplineCutoff, DistanceCutoff, DistanceNeighborFinder, EwaldExclusions, GPUNeighborFinder,
    HarmonicAngles, HarmonicBonds, Langevin, LennardJones, NeighborList, NoCutoff, NoNeighborFinder, PeriodicTorsions,
    PME, PolynomialCutoff, ShiftedForceCutoff, ShiftedPotentialCutoff, System, TriclinicBoundary, VelocityVerlet, find_neighbors, forces,
    kinetic_energy, potential_energy, pressure, scalar_pressure, random_velocities, apply_coupling, remove_CM_motion, scalar_virial, simulate, temperature, total_energy, use_neighbors, virial,
    wrap_coords, optimize_launch_config, set_launch_config, MonteCarloBarostat, SteepestDescentMinimizer, scale_boundary, scale_coords, volume, BAR,
)
