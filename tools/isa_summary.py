"""Scratch: summarise kernels of a hipcc -S device listing: registers, spills, and the instruction mix between barriers.
usage: isa_summary.py file.s <substring of mangled name> [--dump]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
dump = '--dump' in sys.argv
i = 0
while i < len(src):
    l = src[i]
    mm = re.match(r'^(_Z\S+):', l)
    if mm and pat in mm.group(1):
        name = mm.group(1)
        j = i + 1
        body = []
        while j < len(src) and not src[j].startswith('\t.end_amdhsa_kernel') and not (src[j].startswith('.Lfunc_end')):
            body.append(src[j]); j += 1
        # resource lines follow
        meta = {}
        k = j
        while k < len(src) and k < j + 400:
            m = re.match(r'\s*;\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|codeLenInByte)\s*[:=]?\s*(\d+)', src[k])
            if m: meta[m.group(1)] = int(m.group(2))
            if 'Occupancy' in meta and 'LDSByteSize' in meta: break
            k += 1
        print(name[:110]); print('  ', meta)
        # instruction mix per segment between s_barrier
        seg = {}; segs = []
        for b in body:
            t = b.strip().split()
            if not t or t[0].startswith((';', '.')) : continue
            op = t[0]
            if op == 's_barrier':
                segs.append(seg); seg = {}
                continue
            key = None
            if op.startswith('v_mfma'): key = 'mfma'
            elif op.startswith('ds_read') or op.startswith('ds_load'): key = 'ds_read'
            elif op.startswith('ds_write') or op.startswith('ds_store'): key = 'ds_write'
            elif 'load_lds' in op or (op.startswith(('global_load', 'buffer_load')) and ' lds' in b): key = 'glds'
            elif op.startswith(('global_load', 'buffer_load')): key = 'gload'
            elif op.startswith(('global_store', 'buffer_store')): key = 'gstore'
            elif op == 's_waitcnt': key = 'wait:' + ' '.join(t[1:])
            elif op.startswith('scratch_'): key = 'scratch'
            elif op.startswith('s_cbranch'): key = 'branch'
            elif op.startswith('v_'): key = 'valu'
            elif op.startswith('s_'): key = 'salu'
            if key: seg[key] = seg.get(key, 0) + 1
        segs.append(seg)
        for n, sg in enumerate(segs):
            print('   seg', n, {k: v for k, v in sorted(sg.items())})
        if dump:
            print('\n'.join(body))
        i = j
    i += 1
