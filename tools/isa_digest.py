#!/usr/bin/env python3
"""Per-kernel digests of the gfx950 machine code in the built objects (fewbit_amd/csrc/build/gfx950/*.o): the device code object
is taken out of each object's .hip_fatbin, disassembled with llvm-objdump, and every kernel's instruction text is hashed
(addresses and encodings dropped), once in program order and once sorted (the instruction multiset).  Two builds whose digests
agree run the same instructions: how a source clean-up is shown to change no bit of output.

    python3 tools/isa_digest.py [build dir] > digests.txt         # one line per kernel: <sha256[:16] in order> <sha256[:16] sorted> <instructions> <symbol>
    python3 tools/isa_digest.py --diff old.txt new.txt            # what differs (kernels added / removed / changed)
"""
import hashlib
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path('/opt/rocm/lib/llvm/bin')
ROOT = Path(__file__).resolve().parents[1]


def digests(build: Path):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(build.glob('*.o')):
            fat, co = Path(tmp) / (obj.stem + '.fatbin'), Path(tmp) / (obj.stem + '.co')
            subprocess.run([LLVM / 'llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat], check=True)
            subprocess.run([LLVM / 'clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}', f'--output={co}', '--unbundle'], check=True)
            text = subprocess.run([LLVM / 'llvm-objdump', '-d', co], check=True, capture_output=True, text=True).stdout
            name, body = None, []

            def flush():
                if name is not None:
                    out[name] = (hashlib.sha256('\n'.join(body).encode()).hexdigest()[:16], hashlib.sha256('\n'.join(sorted(body)).encode()).hexdigest()[:16], len(body))
            for line in text.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    flush()
                    name, body = m.group(1), []
                elif name is not None and line.startswith('\t'):
                    body.append(re.sub(r'\s*//.*$', '', line).strip())
            flush()
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--diff':
        a, b = ({l.split(' ', 3)[3].strip(): l.split(' ', 3)[:3] for l in open(f) if l.strip() and not l.startswith('#')} for f in sys.argv[2:4])
        changed = [k for k in a if k in b and a[k][0] != b[k][0]]
        reordered = [k for k in changed if a[k][1] == b[k][1]]
        print(f'{len(a)} device functions before, {len(b)} after; removed {len(a.keys() - b.keys())}, added {len(b.keys() - a.keys())}, identical {len([k for k in a if k in b]) - len(changed)}, '
              f'same instructions in another order {len(reordered)}, different instructions {len(changed) - len(reordered)}')
        for k in sorted(a.keys() - b.keys()):
            print('removed', k)
        for k in sorted(b.keys() - a.keys()):
            print('added  ', k)
        for k in changed:
            print('reordered' if k in reordered else 'CHANGED  ', a[k], '->', b[k], k)
        return 1 if len(changed) > len(reordered) else 0
    build = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / 'fewbit_amd' / 'csrc' / 'build' / 'gfx950'
    d = digests(build)
    print(f'# {len(d)} device functions in {build}')
    for k, (h, hs, n) in sorted(d.items()):
        print(h, hs, n, k)
    return 0


if __name__ == '__main__':
    sys.exit(main())
