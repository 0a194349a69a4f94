#!/bin/bash
# ffn2_bisect.sh — VERDICT r5 item 4(a): ISA-level bisect of the stand-alone 16x16x32 reproducer (profiles/scripts/ubench/ffn2_coresidency.hip).
# Every variant is ONE change to the 16x16x32 build of the library's own ffn2_kernel; each runs alone / beside VALU-only busy waves / beside a wave that issues
# 16x16x32 matrix instructions, `ROUNDS` runs per cell.  Second table: the base build at row counts around the chip's resident capacity (2 workgroups x 256 CUs x 128
# rows = 65 536 rows): does a difference need waves that are LAUNCHED while other waves are in flight?
# Usage (GPU box):  bash profiles/scripts/ffn2_bisect.sh gpurun_out/r06_coresidency        (builds were made in the container: gpurun_out/ffn2_bisect/bin/*)
set -u
OUT=${1:-gpurun_out/r06_coresidency}
BIN=${2:-tests/tmp/ffn2_bisect}
ROUNDS=${ROUNDS:-20}
mkdir -p "$OUT"
T="$OUT/ffn2_bisect_table.txt"
: > "$T"
for v in base pad100 nop48 vgprform agprform noovfl pair pair_nop48; do
    [ -x "$BIN/ffn2_$v" ] || { echo "missing $BIN/ffn2_$v" >> "$T"; continue; }
    echo "== variant $v" >> "$T"
    timeout 300 "$BIN/ffn2_$v" "$ROUNDS" >> "$T" 2>&1
done
R="$OUT/ffn2_rows_table.txt"
: > "$R"
for rows in 16384 32768 61440 65536 66560 69632 81920 102400 131072 204800; do
    echo "== base (16x16x32), rows $rows" >> "$R"
    timeout 300 "$BIN/ffn2_base" "$ROUNDS" "$rows" 2>&1 | grep -v "^ffn2_kernel" >> "$R"
done
echo done
