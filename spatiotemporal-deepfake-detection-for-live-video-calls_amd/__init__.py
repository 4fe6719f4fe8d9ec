"""MI355X-native AltFreezing (I3D-ResNet-50) clip classifier: the one hot path of
Mariachiar/Spatiotemporal-Deepfake-Detection-for-Live-Video-Calls, rebuilt as hand-written
HIP (gfx950) behind the reference's classifier-plugin surface.  See DESIGN.md."""
from . import arch, synth  # noqa: F401
