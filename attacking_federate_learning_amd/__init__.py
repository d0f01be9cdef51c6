"""MI355X-native Byzantine-robust aggregation engine: the defences.py / malicious.py hot path of
shaneson0/attacking_federate_learning on hand-written HIP kernels (gfx950), behind the reference's own
module names and call signatures.  See DESIGN.md and INTEGRATION.md."""
from . import _native  # noqa: F401

__all__ = ['defences', 'malicious', 'engine', 'sharded']
__version__ = '0.1.0'
