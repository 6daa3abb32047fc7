python - <<'P'
import sys; sys.path.insert(0,'.')
from famsa_amd import seqio
c,o=seqio.synth_uniform(100000,400); seqio.to_fasta(c,o,'/tmp/cmp_100k.fasta')
P
for gt in sl slink upgma slink upgma; do famsa_amd/famsa-gpu -v -gt $gt -gt_export /tmp/cmp_100k.fasta /tmp/o.dnd 2>&1 | tr '\n' ' ' | sed 's/time\.//g'; echo " [$gt]"; done
