"""ctypes binding of the C-ABI compute back end (include/trhip.h -> lib/libtrhip.so).

Thin and literal: one Python method per C entry point, numpy arrays in and out.  There is NO CPU
fallback: if the HIP library is missing or no GPU is visible this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRHIP_LIB") or os.path.join(_HERE, "lib", "libtrhip.so")   # TRHIP_LIB: experiment builds only

FORMAT_R16_FLOAT = 1
FORMAT_R32_FLOAT = 2

BIND_CONSTANT_BUFFER, BIND_PUSH_CONSTANTS, BIND_STRUCTURED_SRV, BIND_STRUCTURED_UAV, BIND_TEXTURE_SRV, BIND_TEXTURE_UAV, BIND_SAMPLER = range(7)

# every symbol include/trhip.h declares (checked by tests/test_abi_symbols.py)
ABI_SYMBOLS = [
    "trhip_last_error", "trhip_abi_version", "trhip_shader_count", "trhip_shader_name", "trhip_shader_exists",
    "trhip_device_create", "trhip_device_create_on_stream", "trhip_device_destroy", "trhip_device_wait_idle",
    "trhip_device_info", "trhip_device_stream", "trhip_device_join_side_stream",
    "trhip_heap_create", "trhip_heap_release",
    "trhip_buffer_create", "trhip_buffer_wrap", "trhip_buffer_memory_requirements", "trhip_buffer_bind_memory",
    "trhip_buffer_retain", "trhip_buffer_release", "trhip_buffer_device_ptr", "trhip_buffer_size",
    "trhip_texture_create", "trhip_texture_memory_requirements", "trhip_texture_bind_memory", "trhip_texture_retain",
    "trhip_texture_release", "trhip_texture_device_ptr", "trhip_texture_mip_info", "trhip_texture_size",
    "trhip_buffer_upload", "trhip_buffer_download", "trhip_texture_upload", "trhip_texture_download",
    "trhip_buffer_mark_written", "trhip_texture_mark_written",
    "trhip_cmd_create", "trhip_cmd_release", "trhip_cmd_open", "trhip_cmd_close", "trhip_cmd_write_buffer",
    "trhip_cmd_clear_buffer_u32", "trhip_cmd_clear_texture_f32", "trhip_cmd_copy_buffer", "trhip_cmd_copy_texture", "trhip_cmd_host_callback", "trhip_cmd_dispatch", "trhip_cmd_dispatch_indirect",
    "trhip_cmd_begin_timer", "trhip_cmd_end_timer", "trhip_cmd_begin_marker", "trhip_cmd_end_marker",
    "trhip_queue_execute",
    "trhip_timer_create", "trhip_timer_release", "trhip_timer_get_ms",
    "trhip_profile_enable", "trhip_profile_filter", "trhip_profile_reset", "trhip_profile_count", "trhip_profile_entry",
    "trhip_launch_shard_late_info",
    "trhip_stream_create", "trhip_stream_create_priority", "trhip_stream_destroy", "trhip_stream_synchronize", "trhip_event_create", "trhip_event_destroy",
    "trhip_event_record", "trhip_stream_wait_event",
]

HOST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)       # trhip_host_fn(user, hip_stream)


class BufferDesc(C.Structure):
    _fields_ = [("byteSize", C.c_uint64), ("structStride", C.c_uint32), ("canHaveUAVs", C.c_uint32),
                ("isDrawIndirectArgs", C.c_uint32), ("isVirtual", C.c_uint32), ("isVolatileConstant", C.c_uint32),
                ("debugName", C.c_char_p)]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mipLevels", C.c_uint32), ("format", C.c_uint32),
                ("isUAV", C.c_uint32), ("isVirtual", C.c_uint32), ("debugName", C.c_char_p)]


class Binding(C.Structure):
    _fields_ = [("type", C.c_uint32), ("slot", C.c_uint32), ("resource", C.c_void_p), ("baseMip", C.c_uint32),
                ("reserved", C.c_uint32)]


class TrhipError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load lib/libtrhip.so; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TrhipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for the HIP back end)")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.trhip_last_error.restype = C.c_char_p
    L.trhip_abi_version.restype = u32
    L.trhip_shader_count.restype = u32
    L.trhip_shader_name.restype = C.c_char_p
    L.trhip_shader_name.argtypes = [u32]
    L.trhip_shader_exists.argtypes = [C.c_char_p]
    L.trhip_device_create.argtypes = [i32, C.POINTER(vp)]
    L.trhip_device_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.trhip_device_destroy.argtypes = [vp]
    L.trhip_device_destroy.restype = None
    L.trhip_device_wait_idle.argtypes = [vp]
    L.trhip_device_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
    L.trhip_device_stream.argtypes = [vp]
    L.trhip_device_stream.restype = vp
    L.trhip_heap_create.argtypes = [vp, u64, C.POINTER(vp)]
    L.trhip_heap_release.argtypes = [vp]
    L.trhip_heap_release.restype = None
    L.trhip_buffer_create.argtypes = [vp, C.POINTER(BufferDesc), C.POINTER(vp)]
    L.trhip_buffer_wrap.argtypes = [vp, vp, C.POINTER(BufferDesc), C.POINTER(vp)]
    L.trhip_buffer_memory_requirements.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.trhip_buffer_bind_memory.argtypes = [vp, vp, u64]
    for n in ("trhip_buffer_retain", "trhip_buffer_release", "trhip_texture_retain", "trhip_texture_release",
              "trhip_cmd_release", "trhip_timer_release"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = None
    L.trhip_buffer_device_ptr.argtypes = [vp]
    L.trhip_buffer_device_ptr.restype = vp
    L.trhip_buffer_size.argtypes = [vp]
    L.trhip_buffer_size.restype = u64
    L.trhip_texture_create.argtypes = [vp, C.POINTER(TextureDesc), C.POINTER(vp)]
    L.trhip_texture_memory_requirements.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.trhip_texture_bind_memory.argtypes = [vp, vp, u64]
    L.trhip_texture_device_ptr.argtypes = [vp]
    L.trhip_texture_device_ptr.restype = vp
    L.trhip_texture_mip_info.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
    L.trhip_texture_size.argtypes = [vp]
    L.trhip_texture_size.restype = u64
    L.trhip_buffer_upload.argtypes = [vp, u64, vp, u64]
    L.trhip_buffer_download.argtypes = [vp, u64, vp, u64]
    L.trhip_texture_upload.argtypes = [vp, u32, vp, u64]
    L.trhip_texture_download.argtypes = [vp, u32, vp, u64]
    L.trhip_cmd_create.argtypes = [vp, C.POINTER(vp)]
    L.trhip_cmd_open.argtypes = [vp]
    L.trhip_cmd_close.argtypes = [vp]
    L.trhip_cmd_write_buffer.argtypes = [vp, vp, u64, vp, u64]
    L.trhip_cmd_clear_buffer_u32.argtypes = [vp, vp, u32]
    L.trhip_cmd_clear_texture_f32.argtypes = [vp, vp, C.c_float]
    L.trhip_cmd_copy_buffer.argtypes = [vp, vp, u64, vp, u64, u64]
    L.trhip_cmd_copy_texture.argtypes = [vp, vp, vp]
    L.trhip_cmd_host_callback.argtypes = [vp, HOST_FN, vp]
    L.trhip_launch_shard_late_info.argtypes = [vp, vp, u32, u32, vp]
    L.trhip_stream_create.argtypes = [i32, C.POINTER(vp)]
    L.trhip_stream_create_priority.argtypes = [i32, i32, C.POINTER(vp)]
    L.trhip_stream_destroy.argtypes = [vp]
    L.trhip_stream_destroy.restype = None
    L.trhip_stream_synchronize.argtypes = [vp]
    L.trhip_event_create.argtypes = [i32, C.POINTER(vp)]
    L.trhip_event_destroy.argtypes = [vp]
    L.trhip_event_destroy.restype = None
    L.trhip_event_record.argtypes = [vp, vp]
    L.trhip_stream_wait_event.argtypes = [vp, vp]
    L.trhip_cmd_dispatch.argtypes = [vp, C.c_char_p, C.POINTER(Binding), u32, vp, u32, u32, u32, u32]
    L.trhip_cmd_dispatch_indirect.argtypes = [vp, C.c_char_p, C.POINTER(Binding), u32, vp, u32, vp, u32]
    L.trhip_cmd_begin_timer.argtypes = [vp, vp]
    L.trhip_cmd_end_timer.argtypes = [vp, vp]
    L.trhip_cmd_begin_marker.argtypes = [vp, C.c_char_p]
    L.trhip_cmd_end_marker.argtypes = [vp]
    L.trhip_queue_execute.argtypes = [vp, C.POINTER(vp), u32]
    L.trhip_timer_create.argtypes = [vp, C.POINTER(vp)]
    L.trhip_timer_get_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.trhip_profile_enable.argtypes = [vp, i32]
    L.trhip_profile_filter.argtypes = [vp, C.c_char_p]
    L.trhip_profile_reset.argtypes = [vp]
    L.trhip_buffer_mark_written.argtypes = [vp]
    L.trhip_texture_mark_written.argtypes = [vp]
    L.trhip_profile_count.argtypes = [vp, C.POINTER(u32)]
    L.trhip_profile_entry.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double)]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise TrhipError(f"trhip error {rc}: {load().trhip_last_error().decode(errors='replace')}")


def shader_names():
    L = load()
    return [L.trhip_shader_name(i).decode() for i in range(L.trhip_shader_count())]


class Buffer:
    def __init__(self, dev: "Device", handle, size: int, name: str):
        self.dev, self.h, self.size, self.name = dev, handle, size, name

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        _check(load().trhip_buffer_upload(self.h, offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype=np.uint32, count: int | None = None, offset: int = 0) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = (self.size - offset) // dtype.itemsize if count is None else count
        out = np.empty(n, dtype)
        if n:
            _check(load().trhip_buffer_download(self.h, offset, out.ctypes.data, out.nbytes))
        return out

    @property
    def ptr(self) -> int:
        return load().trhip_buffer_device_ptr(self.h) or 0

    def mark_written(self):
        """The memory was written behind the back end's back (raw pointer, another wrap, an aliased resource): derived data
        (instance cull cache, meshlet cull stream) must be rebuilt.  include/trhip.h: trhip_buffer_mark_written."""
        _check(load().trhip_buffer_mark_written(self.h))

    def release(self):
        if self.h:
            load().trhip_buffer_release(self.h)
            self.h = None


class Texture:
    def __init__(self, dev: "Device", handle, w: int, h: int, mips: int, fmt: int, name: str):
        self.dev, self.h, self.w, self.hgt, self.mips, self.format, self.name = dev, handle, w, h, mips, fmt, name

    def mip_dims(self, k: int):
        return max(self.w >> k, 1), max(self.hgt >> k, 1)

    def _dtype(self):
        return np.uint16 if self.format == FORMAT_R16_FLOAT else np.float32

    def upload_mip(self, k: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, self._dtype())
        _check(load().trhip_texture_upload(self.h, k, arr.ctypes.data, arr.nbytes))

    def download_mip(self, k: int) -> np.ndarray:
        mw, mh = self.mip_dims(k)
        out = np.empty((mh, mw), self._dtype())
        _check(load().trhip_texture_download(self.h, k, out.ctypes.data, out.nbytes))
        return out

    def upload_chain(self, texels: np.ndarray, offsets):
        """texels: packed mip chain (oracle / interop.hzb_layout order)."""
        for k in range(self.mips):
            mw, mh = self.mip_dims(k)
            self.upload_mip(k, texels[offsets[k]:offsets[k] + mw * mh])

    def download_chain(self) -> np.ndarray:
        return np.concatenate([self.download_mip(k).ravel() for k in range(self.mips)])

    def mark_written(self):
        """See Buffer.mark_written: the footprint-min table of an HZB written through its raw pointer is stale."""
        _check(load().trhip_texture_mark_written(self.h))

    def release(self):
        if self.h:
            load().trhip_texture_release(self.h)
            self.h = None


def bind(kind: int, slot: int, res=None, base_mip: int = 0) -> Binding:
    b = Binding()
    b.type, b.slot, b.baseMip = kind, slot, base_mip
    b.resource = res.h if res is not None else None
    return b


def CB(slot, buf): return bind(BIND_CONSTANT_BUFFER, slot, buf)
def PUSH(slot): return bind(BIND_PUSH_CONSTANTS, slot)
def SRV(slot, buf): return bind(BIND_STRUCTURED_SRV, slot, buf)
def UAV(slot, buf): return bind(BIND_STRUCTURED_UAV, slot, buf)
def TEX_SRV(slot, tex): return bind(BIND_TEXTURE_SRV, slot, tex)
def TEX_UAV(slot, tex, mip=0): return bind(BIND_TEXTURE_UAV, slot, tex, mip)
def SAMPLER(slot): return bind(BIND_SAMPLER, slot)


class Timer:
    def __init__(self, dev, handle):
        self.dev, self.h = dev, handle

    def ms(self) -> float:
        v = C.c_float(0)
        _check(load().trhip_timer_get_ms(self.h, C.byref(v)))
        return float(v.value)

    def release(self):
        if self.h:
            load().trhip_timer_release(self.h)
            self.h = None


class CommandList:
    def __init__(self, dev: "Device", handle):
        self.dev, self.h = dev, handle
        self._keep = []
        self._keep_cb = []

    def open(self):
        self._keep.clear()
        self._keep_cb = []
        _check(load().trhip_cmd_open(self.h))
        return self

    def close(self):
        _check(load().trhip_cmd_close(self.h))
        return self

    def write_buffer(self, buf: Buffer, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        _check(load().trhip_cmd_write_buffer(self.h, buf.h, offset, arr.ctypes.data, arr.nbytes))

    def clear_buffer_u32(self, buf: Buffer, value: int = 0):
        _check(load().trhip_cmd_clear_buffer_u32(self.h, buf.h, value))

    def clear_texture_f32(self, tex: Texture, value: float):
        _check(load().trhip_cmd_clear_texture_f32(self.h, tex.h, value))

    def copy_buffer(self, dst: Buffer, src: Buffer, nbytes: int, dst_offset: int = 0, src_offset: int = 0):
        _check(load().trhip_cmd_copy_buffer(self.h, dst.h, dst_offset, src.h, src_offset, nbytes))

    def copy_texture(self, dst: Texture, src: Texture):
        _check(load().trhip_cmd_copy_texture(self.h, dst.h, src.h))

    def constant_buffer(self, data: np.ndarray, name="cb") -> Buffer:
        """Graphic::CreateConstantBuffer (Graphic.h:66-72): volatile CB + writeBuffer."""
        data = np.ascontiguousarray(data)
        cb = self.dev.create_buffer(data.nbytes, name=name, volatile_constant=True)
        self.write_buffer(cb, data)
        self._keep.append(cb)
        return cb

    def dispatch(self, shader: str, bindings, groups, push: np.ndarray | None = None):
        arr = (Binding * len(bindings))(*bindings)
        p, pb = (None, 0) if push is None else (np.ascontiguousarray(push).ctypes.data, np.ascontiguousarray(push).nbytes)
        gx, gy, gz = groups
        _check(load().trhip_cmd_dispatch(self.h, shader.encode(), arr, len(bindings), p, pb, gx, gy, gz))

    def dispatch_indirect(self, shader: str, bindings, args: Buffer, offset: int = 0, push: np.ndarray | None = None):
        arr = (Binding * len(bindings))(*bindings)
        p, pb = (None, 0) if push is None else (np.ascontiguousarray(push).ctypes.data, np.ascontiguousarray(push).nbytes)
        _check(load().trhip_cmd_dispatch_indirect(self.h, shader.encode(), arr, len(bindings), p, pb, args.h, offset))

    def host_callback(self, fn):
        """fn(hip_stream: int) is called while the list is executed, in order (include/trhip.h)."""
        cb = HOST_FN(lambda _user, stream: fn(int(stream or 0)))
        self._keep_cb.append(cb)
        _check(load().trhip_cmd_host_callback(self.h, cb, None))

    def begin_timer(self, t: Timer): _check(load().trhip_cmd_begin_timer(self.h, t.h))
    def end_timer(self, t: Timer): _check(load().trhip_cmd_end_timer(self.h, t.h))
    def begin_marker(self, name: str): _check(load().trhip_cmd_begin_marker(self.h, name.encode()))
    def end_marker(self): _check(load().trhip_cmd_end_marker(self.h))

    def release(self):
        if self.h:
            load().trhip_cmd_release(self.h)
            self.h = None
        for b in self._keep:
            b.release()
        self._keep.clear()


class Device:
    def __init__(self, index: int = 0, stream: int | None = None, handle: int | None = None):
        """handle: wrap an existing trhip_device (e.g. the host library's, trhost_device()); not owned."""
        L = load()
        h = C.c_void_p()
        self.owned = handle is None
        if handle is not None:
            h = C.c_void_p(handle)
        elif stream is None:
            _check(L.trhip_device_create(index, C.byref(h)))
        else:
            _check(L.trhip_device_create_on_stream(index, C.c_void_p(stream), C.byref(h)))
        self.h = h
        cu, wave, mem = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(L.trhip_device_info(self.h, C.byref(cu), C.byref(wave), C.byref(mem)))
        self.compute_units, self.wave_size, self.total_mem = cu.value, wave.value, mem.value

    def wait_idle(self):
        _check(load().trhip_device_wait_idle(self.h))

    def create_buffer(self, nbytes: int, name="", stride=4, uav=True, indirect=False, virtual=False, volatile_constant=False) -> Buffer:
        d = BufferDesc(int(nbytes), stride, int(uav), int(indirect), int(virtual), int(volatile_constant), name.encode())
        h = C.c_void_p()
        _check(load().trhip_buffer_create(self.h, C.byref(d), C.byref(h)))
        return Buffer(self, h, int(nbytes), name)

    def wrap_buffer(self, ptr: int, nbytes: int, name="", stride=4, uav=True) -> Buffer:
        d = BufferDesc(int(nbytes), stride, int(uav), 0, 0, 0, name.encode())
        h = C.c_void_p()
        _check(load().trhip_buffer_wrap(self.h, C.c_void_p(ptr), C.byref(d), C.byref(h)))
        return Buffer(self, h, int(nbytes), name)

    def buffer_from(self, arr: np.ndarray, name="", uav=True, min_bytes: int = 4) -> Buffer:
        arr = np.ascontiguousarray(arr)
        b = self.create_buffer(max(arr.nbytes, min_bytes), name=name, stride=max(arr.dtype.itemsize, 1), uav=uav)
        if arr.nbytes:
            b.upload(arr)
        return b

    def create_texture(self, w: int, h: int, mips: int, fmt: int, name="", uav=True) -> Texture:
        d = TextureDesc(w, h, mips, fmt, int(uav), 0, name.encode())
        hd = C.c_void_p()
        _check(load().trhip_texture_create(self.h, C.byref(d), C.byref(hd)))
        return Texture(self, hd, w, h, mips, fmt, name)

    def create_command_list(self) -> CommandList:
        h = C.c_void_p()
        _check(load().trhip_cmd_create(self.h, C.byref(h)))
        return CommandList(self, h)

    def create_timer(self) -> Timer:
        h = C.c_void_p()
        _check(load().trhip_timer_create(self.h, C.byref(h)))
        return Timer(self, h)

    def execute(self, *lists: CommandList):
        arr = (C.c_void_p * len(lists))(*[cl.h for cl in lists])
        _check(load().trhip_queue_execute(self.h, arr, len(lists)))

    def profile_enable(self, on: bool = True): _check(load().trhip_profile_enable(self.h, int(on)))
    def profile_reset(self): _check(load().trhip_profile_reset(self.h))
    def profile_filter(self, name: str | None): _check(load().trhip_profile_filter(self.h, name.encode() if name else None))

    def profile(self) -> dict:
        n = C.c_uint32()
        _check(load().trhip_profile_count(self.h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name, cnt, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            _check(load().trhip_profile_entry(self.h, i, C.byref(name), C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = (int(cnt.value), float(ms.value))
        return out

    def destroy(self):
        if self.h and self.owned:
            load().trhip_device_destroy(self.h)
        self.h = None
