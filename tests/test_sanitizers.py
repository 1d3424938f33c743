"""Host sanitizers over the threaded gate codec (csrc/gatestream.hip is host C++: decoder threads, arenas, a shared dictionary, the encoder's
ring of buffers).  GPU AddressSanitizer does not exist on the pool; this is the part of the library a sanitizer CAN see: the file is compiled
host-only with -fsanitize=thread and -fsanitize=address,undefined, linked to tools/sanitize/gatestream_harness.cpp (encode raw / brotli x
copies, decode on 1 and 6 threads, compare with the source, truncated / corrupted / mis-sized streams) and must finish without a report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('sanitizer', ['address', 'thread'])
def test_gate_codec_under_sanitizer(sanitizer):
    cxx = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(cxx) or not shutil.which('bash'):
        pytest.skip('no clang with sanitizer runtimes here')
    p = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'sanitize', 'run.sh'), sanitizer], capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    if p.returncode != 0 and ('cannot find' in tail or 'unsupported option' in tail or 'libclang_rt' in tail):
        pytest.skip('sanitizer runtime not installed: ' + tail[-300:])
    assert p.returncode == 0 and 'sanitizers: clean' in p.stdout, tail
    assert 'WARNING: ThreadSanitizer' not in tail and 'ERROR: AddressSanitizer' not in tail and 'runtime error' not in tail
