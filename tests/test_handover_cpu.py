"""Hand-over of device memory between torch and the library, checked without a GPU: a stubbed
torch.cuda and a recording stand-in for libck_hip.so.  What must hold structurally:
  * every device tensor crosses the C-ABI through Context._in / Context._out, which order the context's stream behind
    torch's current stream (ck_stream_wait) -- for a fresh output AFTER its allocation;
  * the module-level _in refuses a device tensor (a new call site cannot bypass the ordering by accident);
  * pipeline._take keeps its index tensor alive until its gather has run;
  * Context.close() keeps the handle of a context it could not free."""
import ctypes as C
import types

import numpy as np
import pytest

from camkifu_amd import capi, pipeline


class _FakeStream:
    def __init__(self, log, handle=0x5157):
        self.log, self.cuda_stream = log, handle

    def synchronize(self):
        self.log.append("stream.synchronize")


class _FakeTensor:
    is_cuda, device = True, "cuda:0"
    __module__ = "torch"

    def __init__(self, log, shape=(2, 4, 4, 3), name="tensor"):
        self.log, self.shape, self.name = log, shape, name

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return 0x1000

    def __del__(self):
        self.log.append("free " + self.name)


_FakeTensor.__module__ = "torch"


class _FakeLib:
    def __init__(self, log, destroy_rc=0):
        self.log, self.destroy_rc = log, destroy_rc

    def ck_stream_wait(self, h, stream):
        self.log.append("ck_stream_wait(0x%x)" % stream.value)
        return 0

    def ck_ctx_destroy2(self, h):
        self.log.append("ck_ctx_destroy2")
        return self.destroy_rc

    def ck_last_error(self, h):
        return b""


@pytest.fixture
def stub(monkeypatch):
    import torch
    log = []
    monkeypatch.setattr(capi, "lib", lambda: _FakeLib(log))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: _FakeStream(log))
    monkeypatch.setattr(torch, "empty", lambda shape, dtype=None, device=None: (log.append("torch.empty"), _FakeTensor(log, shape, "output"))[1])
    ctx = capi.Context.__new__(capi.Context)
    ctx._h, ctx.device = C.c_void_p(1), 0
    yield ctx, log
    ctx._h = C.c_void_p()


def test_a_fresh_output_is_ordered_behind_torch_after_its_allocation(stub):
    ctx, log = stub
    like = _FakeTensor(log, name="input")
    out, ptr, space = ctx._out(like, (2, 19, 19), np.uint8)
    assert space == capi.CK_DEVICE and ptr.value == 0x1000
    assert log[:2] == ["torch.empty", "ck_stream_wait(0x5157)"]          # allocate, THEN order; no host wait
    assert "stream.synchronize" not in log


def test_an_input_is_ordered_behind_torch_without_a_host_wait(stub):
    ctx, log = stub
    ptr, space, keep = ctx._in(_FakeTensor(log, name="input"))
    assert space == capi.CK_DEVICE and log == ["ck_stream_wait(0x5157)"]
    a = np.zeros((4, 4, 3), np.uint8)
    ptr, space, keep = ctx._in(a)                                         # host memory: nothing to order
    assert space == capi.CK_HOST and log.count("ck_stream_wait(0x5157)") == 1 and "stream.synchronize" not in log


def test_the_module_level_helper_refuses_device_memory(stub):
    ctx, log = stub
    with pytest.raises(capi.CkError, match="Context._in"):
        capi._in(_FakeTensor(log))


def test_every_device_pointer_in_capi_goes_through_the_two_helpers():
    """no data_ptr() / torch.empty in capi.py outside Context._in / Context._out_on"""
    import ast
    import inspect
    tree = ast.parse(inspect.getsource(capi))
    offenders = []
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef)]:
            for node in ast.walk(fn):
                hit = isinstance(node, ast.Attribute) and node.attr in ("data_ptr", "empty") and isinstance(node.value, ast.Name)
                if hit and (node.attr == "data_ptr" or node.value.id == "torch"):
                    if (cls.name, fn.name) not in (("Context", "_in"), ("Context", "_out_on")):
                        offenders.append((cls.name, fn.name, node.attr))
    assert not offenders, offenders


def test_take_keeps_its_index_tensor_until_the_gather_has_run(monkeypatch):
    """round 6: no host wait in _take.  The gather is queued on the calling thread's torch stream; _take hands back that
    stream (the consumer -- a lane thread -- orders its context behind it: Context.wait_stream) and parks the index tensor on
    the batch's keep list, which finish() clears once the batch's queued work has run."""
    import torch
    log, keep = [], []

    class Frames:
        device = "cuda:0"

        def index_select(self, dim, index):
            log.append("gather queued")
            return _FakeTensor(log, name="gathered")
    producer = _FakeStream(log, handle=0xBEEF)
    monkeypatch.setattr(torch, "as_tensor", lambda idx, device=None: _FakeTensor(log, name="index"))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: producer)
    out, stream = pipeline._take(Frames(), [0, 3, 4, 9], keep)
    assert stream is producer and len(keep) == 1
    assert log == ["gather queued"], log                                  # no synchronize, the index tensor is alive
    # the consumer: a context ordered behind the producer's stream before the pointer crosses the C-ABI
    lib = _FakeLib(log)
    monkeypatch.setattr(capi, "lib", lambda: lib)
    ctx = capi.Context.__new__(capi.Context)
    ctx._h, ctx.device = C.c_void_p(7), 0
    ctx.wait_stream(stream)
    assert log[-1] == "ck_stream_wait(0xbeef)"
    ctx._h = C.c_void_p()
    keep.clear()                                                          # what finish() does
    assert log[-1] == "free index"
    del out
    # consecutive frames are a view: no gather, nothing to wait for
    log.clear()
    fr = np.arange(10)
    part, stream = pipeline._take(fr, [2, 3, 4])
    assert part.tolist() == [2, 3, 4] and stream is None and log == []


def test_no_host_wait_is_left_in_the_pipeline():
    """the only synchronize() in pipeline.py is the exchange thread's final drain of its own stream"""
    import inspect
    import re
    src = inspect.getsource(pipeline)
    calls = [m.start() for m in re.finditer(r"\.synchronize\(\)", src)]
    assert len(calls) == 1 and "self._xstream.synchronize()" in src


def test_close_keeps_the_handle_of_a_context_it_could_not_free(monkeypatch):
    log = []
    busy = _FakeLib(log, destroy_rc=capi.CK_ERR_STATE if hasattr(capi, "CK_ERR_STATE") else 5)
    monkeypatch.setattr(capi, "lib", lambda: busy)
    ctx = capi.Context.__new__(capi.Context)
    ctx._h, ctx.device = C.c_void_p(7), 0
    with pytest.raises(capi.CkError, match="not freed"):
        ctx.close()
    assert ctx._h.value == 7                                              # still there: a later close() can retry
    busy.destroy_rc = 0
    ctx.close()
    assert not ctx._h and log == ["ck_ctx_destroy2", "ck_ctx_destroy2"]
