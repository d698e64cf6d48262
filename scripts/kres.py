#!/usr/bin/env python
"""Per-kernel register / spill / LDS table from a hipcc -save-temps gfx950 .s file (development aid)."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, lds, name, priv, ss, vg, vs = m.groups()
    print(f"vgpr={vg:>4} agpr={ag:>4} vspill={vs:>4} sspill={ss:>3} scratch={priv:>5} lds={lds:>6}  {name[:110]}")
