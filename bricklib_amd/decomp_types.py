"""ctypes mirrors of the exchange-plan records of include/brk.h (kept apart from decomp.py so that _lib.py can name them
in its prototypes without a circular import)."""
import ctypes as C


class XCopy(C.Structure):
    """brk_xcopy"""
    _fields_ = [("stage", C.c_uint32), ("phase", C.c_uint32), ("src_space", C.c_uint32), ("dst_space", C.c_uint32),
                ("src_peer", C.c_int32), ("dst_peer", C.c_int32), ("src_off", C.c_uint64), ("dst_off", C.c_uint64),
                ("bytes", C.c_uint64)]


class XMsg(C.Structure):
    """brk_xmsg"""
    _fields_ = [("stage", C.c_uint32), ("is_send", C.c_uint32), ("peer", C.c_int32), ("space", C.c_uint32),
                ("off", C.c_uint64), ("bytes", C.c_uint64)]
