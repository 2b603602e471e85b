"""sdpsr-hip: MI355X-native Jordan-reduction path of SDPSymmetryReduction.jl.

The directory name is not an importable dotted name; load it with
``__graft_entry__.load_package()`` (imports it as ``sdpsr_amd``).
"""
from . import parallel  # noqa: F401
from . import _lib  # noqa: F401
from . import api  # noqa: F401
from ._lib import (MEM_DEVICE, MEM_HOST, SQUARE_AUTO, SQUARE_F32, SQUARE_F64, SQUARE_I8,  # noqa: F401
                   load_library)
from .api import (BlockDiagonalization, Context, DimensionMismatch, InvalidDecompositionField,  # noqa: F401
                  LabelOverflow, NotConverged, NumericalInconsistency, Partition, SdpsrError,
                  admissible_setup, admissible_subspace, blockDiagonalize, default_context, desymmetrize, unSymmetrize,
                  diagonalize, dim, eigen_decomposition, jordan_reduce_batch, Problem, eigen_decomposition_batched, fill, partition_checksum, randomize, reduce_constraints,
                  refine, relabel_keys)
