"""Import-name alias: ``import pytorch_points`` (the reference's package name) resolves to
``pytorch_points_amd``, so code written against the reference's hot-path API --

    from pytorch_points.network.model_loss import nndistance
    from pytorch_points.network.operations import QueryAndGroup, gather_points, ball_query
    from pytorch_points.network.geo_operations import furthest_point_sample
    from pytorch_points.network.pointnet2_utils import three_nn, three_interpolate
    from pytorch_points._ext import losses, sampling

-- runs unchanged with this repository on ``sys.path`` (no install call).  This module replaces
itself in ``sys.modules`` with ``pytorch_points_amd`` and registers the sub-modules of the path;
modules of the reference outside the path (utils, misc, the torch-composed losses) do not exist
here and raise ImportError."""
import pytorch_points_amd as _impl

_impl.install_as_pytorch_points()
