#!/usr/bin/env bash
# Launcher for the two training stages on one MI355X node: one process per GPU over RCCL / xGMI.
# Counterpart of the reference's 3d_ldm/train_LDM.sh:71-76 and 3d_ldm/train_stable.sh:54-66 (torchrun --nproc_per_node=N per stage).
# Unlike those scripts it does NOT set NCCL_P2P_DISABLE / NCCL_IB_DISABLE (train_LDM.sh:41-42, train_stable.sh:44-45): on MI355X the
# peer-to-peer path IS the fabric (7 xGMI links per GPU); disabling it would push every gradient bucket through host memory.
#
#   ./train_LDM.sh [-n GPUS] [-c CONFIG] [-e ENVIRONMENT] [-s autoencoder|diffusion|both] [-- extra flags for the stage script]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
GPUS="${GPUS:-8}"
CONFIG="${CONFIG:-$HERE/config/config_synthetic_train.json}"   # pass -c path/to/config_train_16g.json for the reference's schema
ENVIRONMENT="${ENVIRONMENT:-$HERE/config/environment_synthetic_train.json}"
STAGE="both"
while getopts "n:c:e:s:" opt; do
    case "$opt" in
        n) GPUS="$OPTARG" ;;
        c) CONFIG="$OPTARG" ;;
        e) ENVIRONMENT="$OPTARG" ;;
        s) STAGE="$OPTARG" ;;
        *) echo "usage: $0 [-n GPUS] [-c CONFIG] [-e ENVIRONMENT] [-s autoencoder|diffusion|both] [-- extra flags]" >&2; exit 2 ;;
    esac
done
shift $((OPTIND - 1))
[ "${1:-}" = "--" ] && shift

# dmabuf IPC is the only mode the host driver supports (RCCL / device-memory sharing across processes fails without it)
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"
export OMP_NUM_THREADS="${OMP_NUM_THREADS:-4}"          # the reference pins 4 host threads per rank (train_diffusion.py:56)
unset NCCL_P2P_DISABLE NCCL_IB_DISABLE                   # never inherit the reference scripts' settings
PORT="${MASTER_PORT:-29500}"

make -C "$HERE/3d-latent-diffusion-model_amd/csrc" >/dev/null   # libldm3d.so for gfx950 (no-op when it is up to date)

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
    autoencoder) run_stage train_autoencoder.py "$@" ;;
    diffusion)   run_stage train_diffusion.py "$@" ;;
    both)        run_stage train_autoencoder.py "$@" && run_stage train_diffusion.py "$@" ;;
    *) echo "unknown stage: $STAGE" >&2; exit 2 ;;
esac
