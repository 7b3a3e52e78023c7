"""implicit-sdf-planner_amd — MI355X-native collision cost/gradient engine (one hot path of Implicit-SDF-Planner).

Host-side Python mirror of the C-ABI in include/isdf_accel.h.  The compute lives in csrc/ (hand-written HIP for
gfx950, built into lib/libisdf_accel.so); this package only marshals arrays.  Import name: ``isdf_amd``
(the directory name contains a hyphen; __graft_entry__.load_package() registers the alias).
"""
from . import capi, synth, parallel, fixtures  # noqa: F401
from .engine import Engine, IsdfError, lbfgs_minimize  # noqa: F401
