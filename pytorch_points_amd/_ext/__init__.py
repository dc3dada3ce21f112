"""Stand-ins for the reference's pybind modules ``pytorch_points._ext.losses`` and
``pytorch_points._ext.sampling`` (same function names and positional signatures), implemented on
the C ABI of libpp_hip.so.  ``linalg`` (batched SVD) is out of scope."""
from . import losses, sampling  # noqa: F401
