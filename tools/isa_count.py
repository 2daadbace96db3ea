#!/usr/bin/env python3
"""Instruction counts of one kernel in hipcc's assembly output (developer tool).
Usage: tools/isa_count.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys


def kernel_body(txt, key):
    m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S | re.M)
    if not m:
        raise SystemExit(f"no kernel matching {key}")
    return m.group(1), m.group(2)


def main():
    name, body = kernel_body(open(sys.argv[1]).read(), sys.argv[2])
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    vmem = sum(v for k, v in c.items() if k.startswith(('global_load', 'buffer_load', 'flat_load', 'global_store')))
    print(name[:70], 'total', len(ins), 'valu', valu, 'vmem', vmem, 'lds', sum(v for k, v in c.items() if k.startswith('ds_')))
    print('  ', c.most_common(16))


if __name__ == '__main__':
    main()
