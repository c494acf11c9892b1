"""calico_amd — MI355X-native reprojection-error LM refinement (CALICO Stage 5).

The product is the HIP library behind include/calico_rp.h (calico_amd/csrc); this
package only binds it for tests, bench.py and torch.distributed launches."""
from . import capi, synth  # noqa: F401
