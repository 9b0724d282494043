"""Development: run vault_amd.isa_check over a hipcc -S file.  usage: isa_hazard.py file.s [kernel-name-substring]
(the build runs the same check over every kernel file: vault_amd/build.py)"""
import sys
sys.path.insert(0, ".")
from vault_amd.isa_check import sgpr_vmem_hazards

found = sgpr_vmem_hazards(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
print("\n".join(found))
print("hazards:", len(found))
sys.exit(1 if found else 0)
