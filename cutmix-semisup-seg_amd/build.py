"""
In-tree build of libcutmixseg_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python cutmix-semisup-seg_amd/build.py [--force]

The shared object lands next to the sources (cutmix-semisup-seg_amd/csrc/libcutmixseg_hip.so) so that it travels
with the repository snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libcutmixseg_hip.so')
STAMP = os.path.join(CSRC, '.build_stamp')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function', '-Wno-pass-failed',
         '-Wno-unused-result']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest():
    h = hashlib.sha256()
    deps = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp'))
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'cutmixseg.h'))
    for p in deps:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _obj_digest(src):
    """Digest of what one object file depends on: its source, every shared header, the flags."""
    h = hashlib.sha256()
    deps = [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp'))
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'cutmixseg.h'))
    for p in deps:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + '.o'
        objs.append(obj)
        # per-object stamp (git-ignored, next to the object): only sources that changed are recompiled
        ostamp, odig = obj + '.stamp', _obj_digest(src)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd), ostamp, odig))
    for src, p, ostamp, odig in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on {}'.format(src))
        with open(ostamp, 'w') as f:
            f.write(odig)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
