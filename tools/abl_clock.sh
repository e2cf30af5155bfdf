#!/bin/bash
# Where do k_trigemm_sq's cycles and the chip's CLOCK go?  The production kernel and its loop ablations (csrc/abl/: -DBOHIP_ABL=1 no
# LDS-DMA, 3 no DMA + no fragment reads, 7 no DMA + no fragment reads + no barriers; results wrong by design), each with the event
# time of the kernel AND the core clock sampled inside it: time = cycles / clock, and on MI355X the clock is an OUTPUT (power budget).
for v in "" abl/libbohip_abl1.so abl/libbohip_abl3.so abl/libbohip_abl7.so ""; do
  echo "== ${v:-production}"
  LIBV=$v python tools/power_probe.py 4096 2>/dev/null | grep -E "^\(a\)|^\(b\)"
done
