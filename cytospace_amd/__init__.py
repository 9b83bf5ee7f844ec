"""cytospace_amd -- MI355X-native drop-in for CytoSPACE's cell-to-spot linear-assignment hot path.

Host code is Python (like the reference); all arithmetic on the path runs in hand-written HIP
kernels reached through the C ABI in include/cytohip.h via ctypes (cytospace_amd/_lib.py).
"""
__version__ = "0.1.0"
