"""eppm_amd -- MI355X-native EPPM optical flow (hand-written HIP kernels behind a C ABI).

Host-side mirror of the reference's interface for the hot path:

    flow = eppm_amd.EPPM();  flow.init(h, w);  flow.set_data(img1, img2);  u, v = flow.compute_flow()

which are the public methods of ``class bao_flow_patchmatch_multiscale_cuda``
(bao_flow_patchmatch_multiscale_cuda.h:36-44).  Everything computes in libeppm_hip.so
(eppm_amd/lib, built by eppm_amd.build); there is no CPU fallback: loading fails loudly
when the library is missing.
"""
from .build import build, lib_path  # noqa: F401
from ._lib import EppmError, lib, select_library  # noqa: F401
from .api import EPPM, EPPMBatch, Params, host_register, host_unregister, pinned_empty  # noqa: F401
from . import io, stages  # noqa: F401
