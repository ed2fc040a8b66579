"""zdr_amd — MI355X-native differentiable path tracer behind the reference's Python API
(``from zdr import Scene, Camera, float3``; /root/reference/__init__.py:1)."""
from .mathtypes import Camera, float3, float4x4
from .render import Scene

__all__ = ["Scene", "Camera", "float3", "float4x4"]
