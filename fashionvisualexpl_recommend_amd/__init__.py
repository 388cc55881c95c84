"""MI355X-native BPRMF / VBPR training hot path.

Light on import: numpy-only helpers (`synth`, `configs`) can be used without a GPU; `models`, `engine`
need libbprx.so and a ROCm device and raise loudly otherwise.
"""
__all__ = ["configs", "synth"]
