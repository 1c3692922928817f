"""The device-index class of bug guarded without a second GPU.

"cuda" without an index means "the calling thread's current device", a per-thread setting that starts at device 0 in every new
thread.  On a rank with LOCAL_RANK > 0 a worker thread that allocates, launches or asks for "the current stream" before selecting the
rank's GPU works on GPU 0.  The package's rules, checked here on the source (no GPU needed):

  1. threads are created in ONE place -- io_pipeline.start_thread / thread_pool -- and both select the creating thread's device
     before the target runs (exercised below on a fake torch.cuda);
  2. the literal "cuda" appears only in _lib.resolve_device (which turns None / "cuda" into an INDEXED device on the calling
     thread, once) and in ``.type == "cuda"`` tests; constructors and helpers default to ``device=None`` and go through it;
  3. resolve_device itself: indexes with the calling thread's current device, a tensor stands for its own device.
"""
import ast
import glob
import os
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "premvos_amd", "*.py")) + glob.glob(os.path.join(ROOT, "premvos_amd", "*", "*.py")))


def _calls(tree):
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            f = node.func
            name = f.attr if isinstance(f, ast.Attribute) else f.id if isinstance(f, ast.Name) else None
            yield name, node


def test_threads_are_created_only_by_the_device_binding_factory():
    assert len(FILES) > 20
    offenders = []
    for fn in FILES:
        tree = ast.parse(open(fn).read())
        for name, node in _calls(tree):
            if name in ("Thread", "ThreadPoolExecutor", "Process", "Pool", "start_new_thread"):
                outer = [f for f in ast.walk(tree) if isinstance(f, ast.FunctionDef) and f.lineno <= node.lineno <= f.end_lineno]
                top = min(outer, key=lambda f: f.lineno).name if outer else None      # the module-level function it sits in
                where = (os.path.relpath(fn, ROOT), top)
                if where not in (("premvos_amd/io_pipeline.py", "start_thread"), ("premvos_amd/io_pipeline.py", "thread_pool")):
                    offenders.append((where, node.lineno))
    assert not offenders, f"threads created outside io_pipeline.start_thread / thread_pool: {offenders}"


def test_the_literal_cuda_appears_only_where_a_device_gets_its_index():
    offenders = []
    for fn in FILES:
        rel = os.path.relpath(fn, ROOT)
        if rel == "premvos_amd/synth.py":                    # synthetic INPUTS for tests / bench (never allocates on a device)
            continue
        tree = ast.parse(open(fn).read())
        ok_lines = set()
        for node in ast.walk(tree):
            # x.type == "cuda"
            if isinstance(node, ast.Compare) and any(isinstance(c, ast.Constant) and c.value == "cuda" for c in node.comparators):
                ok_lines.add(node.lineno)
            # docstrings
            if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant) and isinstance(node.value.value, str):
                ok_lines |= set(range(node.lineno, node.end_lineno + 1))
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and node.value.startswith("cuda") and node.lineno not in ok_lines:
                fdefs = [f for f in ast.walk(tree) if isinstance(f, ast.FunctionDef) and f.lineno <= node.lineno <= f.end_lineno]
                if rel == "premvos_amd/_lib.py" and fdefs and fdefs[-1].name == "resolve_device":
                    continue
                offenders.append((rel, node.lineno))
    assert not offenders, f'bare "cuda" (= whatever device the calling thread happens to have): {offenders}'


def test_every_device_parameter_defaults_to_none_and_is_resolved():
    """A function that takes ``device`` must not default it to a string; if it allocates with it, it resolves it (or hands it to a
    callee of this package, which does)."""
    bad = []
    for fn in FILES:
        tree = ast.parse(open(fn).read())
        for f in ast.walk(tree):
            if not isinstance(f, ast.FunctionDef):
                continue
            args = f.args.args + f.args.kwonlyargs
            defaults = [None] * (len(f.args.args) - len(f.args.defaults)) + list(f.args.defaults) + list(f.args.kw_defaults)
            for a, d in zip(args, defaults):
                if a.arg == "device" and isinstance(d, ast.Constant) and isinstance(d.value, str) and d.value.startswith("cuda"):
                    bad.append((os.path.relpath(fn, ROOT), f.name))
    assert not bad, bad


def test_factory_threads_select_the_creators_device_before_the_target_runs(monkeypatch):
    import torch
    from premvos_amd import io_pipeline as iop
    state = threading.local()
    log = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: getattr(state, "dev", 0))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: setattr(state, "dev", d))
    torch.cuda.set_device(5)                                   # "this rank's GPU", on the creating thread
    t = iop.start_thread(lambda: log.append(("thread", torch.cuda.current_device())), "premvos-test")
    t.join()
    with iop.thread_pool(2, "premvos-test-pool") as pool:
        log += list(pool.map(lambda i: ("pool", torch.cuda.current_device()), range(4)))
    got = [x for x in iop.prefetch([1, 2, 3], lambda j: torch.cuda.current_device(), workers=2)]
    lanes = list(iop.lanes([1, 2, 3, 4], lambda lane, it: torch.cuda.current_device(), n=2))
    w = iop.Writer(threads=2)
    for _ in range(4):
        w.submit(lambda: log.append(("writer", torch.cuda.current_device())))
    w.close()
    assert {d for _, d in log} == {5} and len(log) == 1 + 4 + 4 and got == [5, 5, 5] and lanes == [5, 5, 5, 5]


def test_resolve_device_indexes_once_on_the_calling_thread(monkeypatch):
    import torch
    from premvos_amd import _lib
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert _lib.resolve_device() == torch.device("cuda", 3) == _lib.resolve_device("cuda") == _lib.resolve_device(torch.device("cuda"))
    assert _lib.resolve_device("cuda:1") == torch.device("cuda", 1) and _lib.resolve_device("cpu") == torch.device("cpu")
    assert _lib.resolve_device(torch.zeros(2)) == torch.device("cpu")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert _lib.resolve_device() == torch.device("cuda")      # host-only: passes through (nothing can allocate there anyway)
