"""rfx: host-side package of the MI355X-native RANSAC-Flow hot path (ctypes over librfx.so)."""
__version__ = "0.5.0"
