"""icafusion_amd — MI355X-native hot path for ICAFusion (two-stream YOLOv5 + DMFF + Detect + NMS)."""
__version__ = "0.1.0"
