#!/usr/bin/env python3
"""Static check of the one hazard the hand-written DPP statements manage themselves:
on gfx950 a VALU write of a VGPR needs two wait states before a DPP read of it, and the
compiler's hazard recogniser does not look inside inline asm.  Disassembles a code object
(or reads a .s) and, for every DPP instruction, walks back over the two preceding wait
states (an instruction = 1, s_nop N = N + 1) and reports any VALU write of the DPP source.

    python tools/check_dpp_hazard.py speaker-diarization_amd/csrc/libspkd_hip.so
"""
import re
import subprocess
import sys

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
BUNDLER = '/opt/rocm/lib/llvm/bin/clang-offload-bundler'


def disassemble(path):
    if path.endswith('.s'):
        return open(path).read()
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, 'gfx950.co')
        # shared libraries carry the device code as an offload bundle
        r = subprocess.run([BUNDLER, '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                            '--input=' + path, '--output=' + co, '--unbundle'],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            co = path
        return subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', co], capture_output=True, text=True,
                              check=True).stdout


def regs(tok):
    """VGPR indices named by an operand token: v12, v[12:13]"""
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    text = disassemble(sys.argv[1])
    bad = 0
    n_dpp = 0
    near_label = 0
    hist = []          # (mnemonic, operands) of the current function, program order
    func = '?'
    for line in text.splitlines():
        line = line.split('//')[0].split(';')[0].rstrip()
        if not line.strip() or line.strip().startswith('.'):
            continue
        m = re.match(r'^[0-9a-f]* ?<([^>]+)>:$', line.strip()) or re.match(r'^([A-Za-z_][\w$.]*):$', line.strip())
        if m or line.rstrip().endswith(':'):
            if m and not m.group(1).startswith(('L', '.L')) and '$' not in m.group(1):
                func = m.group(1)
            hist.append(('<label>', []))
            continue
        parts = line.strip().split(None, 1)
        mnem = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        if ('_dpp' in mnem) or any('row_newbcast' in o or 'row_shr' in o or 'quad_perm' in o for o in ops):
            n_dpp += 1
            # DPP applies to src0 = the operand after the destination
            src = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            states = 0
            k = len(hist) - 1
            while states < 2 and k >= 0:
                pm, pops = hist[k]
                if pm == '<label>':
                    near_label += 1
                    break
                if pm == 's_nop':
                    states += int(pops[0], 0) + 1
                else:
                    if pm.startswith('v_') and pops:
                        if regs(pops[0].split()[0]) & src:
                            bad += 1
                            print('HAZARD in %s: %s %s  <- %s %s' % (func, mnem, ', '.join(ops), pm, ', '.join(pops)))
                    states += 1
                k -= 1
        hist.append((mnem, ops))
    print('%d DPP instructions checked, %d hazards, %d within two wait states of a label' % (n_dpp, bad, near_label))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
