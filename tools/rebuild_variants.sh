#!/bin/bash
# Rebuild every diagnostic / ablation variant whose flags file sits in tools/variants/
# (tools/build_variant.sh wrote it) against the current product sources.  Not part of
# build.sh: the product build ships libblueberry_hip.so and nothing else.
cd "$(dirname "$0")/.."
for f in tools/variants/libabl_*.flags; do
    [ -e "$f" ] || continue
    n=$(basename "$f" .flags); n=${n#libabl_}
    tools/build_variant.sh "$n" $(cat "$f")
done
