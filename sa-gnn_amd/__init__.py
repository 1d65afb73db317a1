"""sa-gnn_amd — SelfGNN's per-time-interval graph propagation + interval fusion on MI355X.

Host side in Python, mirroring the reference's entry points (Params.py / DataHandler.py /
model.py); all arithmetic goes through libsagnn.so (hand-written HIP for gfx950) over the C ABI
in include/sagnn.h. There is no CPU fallback: importing `ops` without the built library raises.
"""
__version__ = "1.1.0"
