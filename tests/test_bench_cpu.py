"""Host-side contract of bench.py that needs no GPU: the bare `python bench.py --gpus N` command starts its own ranks."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench(monkeypatch, argv):
    monkeypatch.setattr(sys, 'argv', argv)
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)              # __name__ != '__main__': nothing launches on import
    return mod


def test_bare_command_reexecs_under_the_launcher(monkeypatch):
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    mod = _load_bench(monkeypatch, ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    seen = {}
    monkeypatch.setattr(os, 'execv', lambda exe, cmd: seen.update(exe=exe, cmd=cmd))
    mod._self_launch()
    cmd = seen['cmd']
    assert seen['exe'] == sys.executable and cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']       # the caller's own arguments, verbatim


def test_no_reexec_for_one_gpu_or_under_a_launcher(monkeypatch):
    called = []
    monkeypatch.setattr(os, 'execv', lambda *a: called.append(a))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    _load_bench(monkeypatch, ['bench.py'])._self_launch()
    _load_bench(monkeypatch, ['bench.py', '--gpus=1'])._self_launch()
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setenv('RANK', '0')
    _load_bench(monkeypatch, ['bench.py', '--gpus', '4'])._self_launch()
    assert called == []
