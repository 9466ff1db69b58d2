import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_launcher = None


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The two-process test of mil_mg needs rank processes that are fresh children of a process WITHOUT GPU state (a process
    # that has initialised the GPU must not start programs).  This process has made no GPU call yet, so the idle launcher
    # is started here; tests/test_zy_two_process_mg.py talks to it over its pipes.
    global _launcher
    if 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or ''):
        import subprocess
        _launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'mg_two_proc.py'), 'launcher'],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.write('quit\n'); _launcher.stdin.flush()
            _launcher.wait(timeout=10)
        except Exception:
            _launcher.kill()
        _launcher = None


@pytest.fixture(scope='session')
def mg_launcher():
    return _launcher


@pytest.fixture(scope='session')
def coracle():
    from oracle import coracle as co
    co.build()
    return co
