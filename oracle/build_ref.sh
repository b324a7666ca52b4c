#!/usr/bin/env bash
# TEST INFRASTRUCTURE (oracle). Builds the REAL reference (rbitr/llm.f90 master) from the sources
# where they lie under /root/reference into oracle/_ref/ -- binaries only, never sources.
#
#   oracle/build_ref.sh                      -> oracle/_ref/llm_ref            (unmodified: TinyLlama dims)
#   oracle/build_ref.sh NAME E H L NH NKV V S -> oracle/_ref/llm_ref_NAME      (dims + logits dump)
#
# The reference hard-codes its model dims as Fortran `parameter`s (llama2.f90:102-108) and
# `transformer` is an internal procedure of `program llama2` (llama2.f90:480), so logits are only
# observable by stream-editing a scratch copy at build time:
#   * the seven dim lines are rewritten (semantics untouched),
#   * `write(9) logits` is inserted after the call at llama2.f90:380 (unit 9 opened after :376),
#     producing logits.bin = n x V little-endian f32 in the working directory.
# The scratch copy lives in a mktemp dir that is deleted before the script exits.
# Flags: -O2. NOT -O0: the GQA slice at llama2.f90:581/591 is non-conforming and flang's -O0
# runtime aborts on it; at -O2 it has the intended semantics.
# Needs /root/reference and amdflang; on the GPU box only the prebuilt binaries are used.
set -euo pipefail
REF=${LLMK_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
FC=${FC:-/opt/rocm/bin/amdflang}
[ -d "$REF" ] || { echo "build_ref: $REF not present (GPU box?) - skipping"; exit 0; }
[ -x "$FC" ] || { echo "build_ref: $FC not found - skipping"; exit 0; }
mkdir -p "$OUT"
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
FFLAGS=${REF_FFLAGS:--O2}

if [ $# -eq 0 ]; then
  # unmodified reference, the Makefile's optimisation level minus gfortran-only -flto
  (cd "$T" && "$FC" -O3 -march=native -ffast-math -funroll-loops \
      "$REF/weight_module.f90" "$REF/read_ggml.f90" "$REF/llama2.f90" -o "$OUT/llm_ref" 2>/dev/null)
  echo "built $OUT/llm_ref"
  exit 0
fi

NAME=$1; E=$2; H=$3; L=$4; NH=$5; NKV=$6; V=$7; S=$8
sed -e "s/emb_dim = 2048/emb_dim = $E/" -e "s/hidden_dim = 5632/hidden_dim = $H/" \
    -e "s/n_layers = 22/n_layers = $L/" -e "s/n_heads = 32/n_heads = $NH/" \
    -e "s/n_kv_heads = 4/n_kv_heads = $NKV/" -e "s/vocab_size = 32000/vocab_size = $V/" \
    -e "s/seq_len = 2048/seq_len = $S/" \
    -e "/^        token = 2\$/a\\        open(unit=9,file='logits.bin',form='unformatted',access='stream',status='replace')" \
    -e "/logits = transformer(token,pos,s,weights)/a\\        write(9) logits" \
    "$REF/llama2.f90" > "$T/main_patched.f90"
grep -q "write(9) logits" "$T/main_patched.f90" || { echo "build_ref: harness patch did not apply"; exit 1; }
(cd "$T" && "$FC" $FFLAGS "$REF/weight_module.f90" "$REF/read_ggml.f90" main_patched.f90 -o "$OUT/llm_ref_$NAME" 2>/dev/null)
echo "built $OUT/llm_ref_$NAME"
