"""Print per-kernel resource usage from hipcc -save-temps assembly (dev helper)."""
import re, sys
s = open(sys.argv[1]).read()
meta = s[s.index('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    def g(k):
        m = re.search(r'\.' + k + r':\s+(\S+)', blk)
        return m.group(1) if m else '-'
    print('%-72s vgpr %-4s agpr %-3s sgpr %-4s lds %-6s scratch %-4s spill %s' % (
        g('name')[:72], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('group_segment_fixed_size'),
        g('private_segment_fixed_size'), g('vgpr_spill_count')))
