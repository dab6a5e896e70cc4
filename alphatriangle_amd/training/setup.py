"""Builds every training component (parity: alphatriangle/training/setup.py:29-246).

Ray start-up is replaced by: picking the self-play engine (HBM-resident GPU actor
on an MI355X, CPU thread workers otherwise), clamping the CPU worker count to the
detected cores minus two, and creating the in-process stats actor.
"""

from __future__ import annotations

import logging
import os
from typing import Any

import torch

from ..config import (
    AlphaTriangleMCTSConfig,
    EnvConfig,
    ModelConfig,
    SystemConfig,
    TrainConfig,
    print_config_info_and_validate,
)
from ..models.network import NeuralNetwork
from ..rl.core.buffer import ExperienceBuffer
from ..rl.core.trainer import Trainer
from ..stats import Serializer, StatsActor, TrieyeConfig
from ..utils.helpers import get_device, set_random_seeds
from .components import TrainingComponents

logger = logging.getLogger(__name__)


def count_parameters(model: torch.nn.Module) -> tuple[int, int]:
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return total, trainable


def _clamp_workers(requested: int) -> int:
    cores = os.cpu_count() or 1
    usable = max(1, cores - 2)
    actual = max(1, min(requested, usable))
    if actual != requested:
        logger.info("Requested %d workers, %d cores detected -> using %d.", requested, cores, actual)
    return actual


def setup_training_components(
    train_config_override: TrainConfig,
    trieye_config_override: TrieyeConfig,
    profile: bool,
    model_config_override: ModelConfig | None = None,
    mcts_config_override: AlphaTriangleMCTSConfig | None = None,
    env_config_override: EnvConfig | None = None,
    system_config_override: SystemConfig | None = None,
    dist_context: Any = None,
) -> tuple[TrainingComponents | None, bool]:
    """Returns ``(components, runtime_started)``; ``(None, started)`` on failure."""
    started = False
    stats_actor: StatsActor | None = None
    try:
        train_config = train_config_override
        trieye_config = trieye_config_override
        env_config = env_config_override or EnvConfig()
        model_config = model_config_override or ModelConfig()
        mcts_config = mcts_config_override or AlphaTriangleMCTSConfig()
        system_config = system_config_override or SystemConfig()

        train_config.NUM_SELF_PLAY_WORKERS = _clamp_workers(train_config.NUM_SELF_PLAY_WORKERS)
        train_config.PROFILE_WORKERS = profile
        print_config_info_and_validate(mcts_config, system_config)

        set_random_seeds(train_config.RANDOM_SEED)
        device = get_device(train_config.DEVICE)
        worker_device = get_device(train_config.WORKER_DEVICE)
        logger.info("Determined Training Device: %s", device)
        logger.info("Determined Worker Device: %s", worker_device)

        stats_actor = StatsActor(trieye_config, name=f"trieye_actor_{trieye_config.run_name}")
        started = True
        stats_actor.attach_log_file()
        logger.info("Stats actor ready (run id %s).", stats_actor.get_mlflow_run_id())

        serializer = Serializer()
        neural_net = NeuralNetwork(model_config, env_config, train_config, device, inference_dtype=system_config.INFERENCE_DTYPE)

        use_gpu_engine = system_config.ENGINE == "gpu" or (system_config.ENGINE == "auto" and device.type == "cuda")
        if use_gpu_engine and neural_net.engine is None:
            # a model the HIP kernels do not tile (see ops/inference.py: engine_supports): the GPU tree still runs,
            # its leaves are evaluated by the PyTorch module
            logger.warning("No HIP inference engine for this model config: GPU self-play evaluates leaves with the PyTorch module.")
        split = dist_context is not None and not dist_context.layout.colocated
        pure_actor = split and not dist_context.is_learner
        pure_learner = split and not dist_context.is_actor
        rank = dist_context.rank if dist_context else 0
        gpu_actor = None
        experience_source = None
        buffer: Any
        if use_gpu_engine:
            if device.type != "cuda":
                raise RuntimeError("SystemConfig.ENGINE='gpu' requires a GPU device")
            from ..rl.core.device_buffer import DeviceReplayBuffer
            from ..rl.self_play.gpu_actor import GpuSelfPlayActor

            buf_cfg = train_config
            if pure_actor:
                # actor ranks keep only an outbox: one move's worth of experiences, shipped every round
                cap = max(int(system_config.SHIP_CAPACITY), system_config.GAMES_PER_GPU * (train_config.N_STEP_RETURNS + 1))
                buf_cfg = train_config.model_copy(
                    update={"BUFFER_CAPACITY": cap, "MIN_BUFFER_SIZE_TO_TRAIN": 1, "BATCH_SIZE": 1, "USE_PER": False}
                )
            buffer = DeviceReplayBuffer(buf_cfg, env_config, model_config, device, system_config)
            if not pure_learner:
                profile_dir = None
                if train_config.PROFILE_WORKERS:
                    import os

                    profile_dir = os.path.join(stats_actor.get_run_base_dir_str(), "profile_data")
                gpu_actor = GpuSelfPlayActor(
                    env_config, model_config, mcts_config, train_config, system_config, neural_net, buffer, device,
                    seed=train_config.RANDOM_SEED + rank * 1_000_003, profile_dir=profile_dir, rank=rank,
                )
        else:
            buffer = ExperienceBuffer(train_config)
        if pure_actor:
            from ..parallel.experience import RecordLayout
            from .split_loop import GpuExperienceSource, HostExperienceSource

            layout = RecordLayout(
                model_config.GRID_INPUT_CHANNELS * env_config.ROWS * env_config.COLS,
                model_config.OTHER_NN_INPUT_FEATURES_DIM,
                env_config.ACTION_DIM,
            )
            if gpu_actor is not None:
                experience_source = GpuExperienceSource(gpu_actor, buffer, layout)
            else:
                from ..rl.self_play.worker import SelfPlayWorker

                worker = SelfPlayWorker(
                    rank, env_config, mcts_config, model_config, train_config, stats_actor=stats_actor,
                    run_base_dir=stats_actor.get_run_base_dir_str(), seed=train_config.RANDOM_SEED + rank,
                    worker_device_str=train_config.WORKER_DEVICE,
                )
                worker.nn_evaluator = neural_net  # published weights land in this rank's network
                experience_source = HostExperienceSource(worker, layout, env_config.ACTION_DIM)

        if dist_context is not None and dist_context.world_size > 1:
            dist_context.warm_up()  # communicator set-up must not fall into the first graph captures
        trainer = None
        if not pure_actor:
            n_sync = 1
            grad_sync = None
            if dist_context is not None and dist_context.world_size > 1:
                n_sync = dist_context.num_learners if split else dist_context.world_size
                grad_sync = dist_context.grad_sync if n_sync > 1 else None
            system_config.apply_learner_gemm_mode()
            trainer = Trainer(
                neural_net,
                train_config,
                env_config,
                autocast_bf16=system_config.learner_autocast(),
                grad_sync=grad_sync,
                world_size=n_sync,
            )
        components = TrainingComponents(
            nn=neural_net,
            buffer=buffer,
            trainer=trainer,
            trieye_actor=stats_actor,
            trieye_config=trieye_config,
            serializer=serializer,
            train_config=train_config,
            env_config=env_config,
            model_config=model_config,
            mcts_config=mcts_config,
            profile_workers=train_config.PROFILE_WORKERS,
            system_config=system_config,
            gpu_actor=gpu_actor,
            dist=dist_context,
            experience_source=experience_source,
        )
        return components, started
    except Exception as exc:
        logger.critical("Error setting up training components: %s", exc, exc_info=True)
        if stats_actor is not None:
            try:
                stats_actor.shutdown(timeout=5.0)
            except Exception:  # pragma: no cover
                pass
        return None, started
