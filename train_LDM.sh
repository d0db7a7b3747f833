#!/usr/bin/env bash
# Launcher for the two training stages on one MI355X node: one process per GPU over RCCL / xGMI.
# Counterpart of the reference's 3d_ldm/train_LDM.sh:71-76 and 3d_ldm/train_stable.sh:54-66 (torchrun --nproc_per_node=N per stage).
# Unlike those scripts it does NOT set NCCL_P2P_DISABLE / NCCL_IB_DISABLE (train_LDM.sh:41-42, train_stable.sh:44-45): on MI355X the
# peer-to-peer path IS the fabric (7 xGMI links per GPU); disabling it would push every gradient bucket through host memory.
#
#   ./train_LDM.sh [-n GPUS] [-c CONFIG] [-e ENVIRONMENT] [-s autoencoder|diffusion|both] [-A 'autoencoder-stage flags'] [-D 'diffusion-stage flags']
#                  [-- flags for the ONE stage selected with -s autoencoder|diffusion]
# The two stage scripts parse strictly and take different options (--amp, --no-images, --precision, --perceptual-weights exist only in the
# autoencoder stage), so with -s both the per-stage flags go through -A / -D; trailing "-- flags" with -s both are refused up front instead
# of failing in argparse after the whole first stage has trained.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
GPUS="${GPUS:-8}"
CONFIG="${CONFIG:-$HERE/config/config_synthetic_train.json}"   # pass -c path/to/config_train_16g.json for the reference's schema
ENVIRONMENT="${ENVIRONMENT:-$HERE/config/environment_synthetic_train.json}"
STAGE="both"
AE_FLAGS=""
DM_FLAGS=""
while getopts "n:c:e:s:A:D:" opt; do
    case "$opt" in
        n) GPUS="$OPTARG" ;;
        c) CONFIG="$OPTARG" ;;
        e) ENVIRONMENT="$OPTARG" ;;
        s) STAGE="$OPTARG" ;;
        A) AE_FLAGS="$OPTARG" ;;
        D) DM_FLAGS="$OPTARG" ;;
        *) echo "usage: $0 [-n GPUS] [-c CONFIG] [-e ENVIRONMENT] [-s autoencoder|diffusion|both] [-A 'flags'] [-D 'flags'] [-- extra flags]" >&2; exit 2 ;;
    esac
done
shift $((OPTIND - 1))
[ "${1:-}" = "--" ] && shift

# dmabuf IPC is the only mode the host driver supports (RCCL / device-memory sharing across processes fails without it)
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"
export OMP_NUM_THREADS="${OMP_NUM_THREADS:-4}"          # the reference pins 4 host threads per rank (train_diffusion.py:56)
unset NCCL_P2P_DISABLE NCCL_IB_DISABLE                   # never inherit the reference scripts' settings
PORT="${MASTER_PORT:-29500}"

if [ "$STAGE" = "both" ] && [ "$#" -gt 0 ]; then
    echo "$0: with -s both pass stage flags as -A '...' (autoencoder) and -D '...' (diffusion): the stages take different options (got: $*)" >&2
    exit 2
fi

# libldm3d.so for gfx950 (no-op when it is up to date); the build log is shown only when the build fails
BUILD_LOG="$(mktemp)"
if ! make -C "$HERE/3d-latent-diffusion-model_amd/csrc" >"$BUILD_LOG" 2>&1; then
    cat "$BUILD_LOG" >&2; rm -f "$BUILD_LOG"
    echo "$0: building libldm3d.so failed" >&2
    exit 1
fi
rm -f "$BUILD_LOG"

run_stage() {   # $1 = script; the rest = its flags
    local script="$1"; shift
    if [ "$GPUS" -gt 1 ]; then
        python3 -m torch.distributed.run --nnodes=1 --nproc-per-node="$GPUS" --master-addr 127.0.0.1 --master-port "$PORT" \
            "$HERE/$script" -c "$CONFIG" -e "$ENVIRONMENT" -g "$GPUS" "$@"
    else
        python3 "$HERE/$script" -c "$CONFIG" -e "$ENVIRONMENT" -g 1 "$@"
    fi
}

case "$STAGE" in
    # shellcheck disable=SC2086  (the per-stage flag strings are word-split on purpose)
    autoencoder) run_stage train_autoencoder.py $AE_FLAGS "$@" ;;
    diffusion)   run_stage train_diffusion.py $DM_FLAGS "$@" ;;
    both)        run_stage train_autoencoder.py $AE_FLAGS && run_stage train_diffusion.py $DM_FLAGS ;;
    *) echo "unknown stage: $STAGE" >&2; exit 2 ;;
esac
