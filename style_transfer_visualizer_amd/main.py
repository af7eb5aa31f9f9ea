"""Top-level orchestration: the production caller of the hot path (reference main.py:20-167).

validate -> seed -> device -> load images -> model/targets/optimizer -> run -> save PNG ->
``input_img.detach().clamp(0, 1)``.  Timelapse video / GIF encoding is presentation and is not
part of this build: any ``FrameSink`` may be injected, otherwise video requests are downgraded
with a warning and the run proceeds like ``--no-video``.
"""
from __future__ import annotations

import threading
from pathlib import Path

import torch

from . import core_model, image_io, optimization, runtime
from .logging_utils import logger
from .type_defs import InputPaths, SaveOptions

# Several images may be in flight in one process (style_transfer_batch): seeding the global generators and drawing
# the start image from them is one step per image, as in a sequential run.
_SETUP_LOCK = threading.Lock()


def style_transfer(paths: InputPaths, config, *, video_writer=None, gif_collector=None) -> torch.Tensor:
    """Run one style transfer; returns the optimised image clamped to [0, 1]."""
    runtime.validate_input_paths(paths.content_path, paths.style_path)
    runtime.validate_parameters(config.video.quality)

    if config.video.final_only:                 # reference main.py:30-33
        config.video.create_video = False
        config.video.create_gif = False
        config.video.save_every = config.optimization.steps + 1
    if (config.video.create_video and video_writer is None) or (config.video.create_gif and gif_collector is None):
        logger.warning("Timelapse encoding is not part of the MI355X build; continuing without video/GIF "
                       "(pass --no-video to silence this, or inject a frame sink).")
        config.video.create_video = video_writer is not None
        config.video.create_gif = gif_collector is not None

    device = runtime.setup_device(config.hardware.device)
    normalize = config.optimization.normalize
    content_img = image_io.load_image_to_tensor(paths.content_path, device, normalize=normalize)
    style_img = image_io.load_image_to_tensor(paths.style_path, device, normalize=normalize)
    with _SETUP_LOCK:
        runtime.setup_random_seed(config.optimization.seed)
        model, input_img, optimizer = core_model.prepare_model_and_input(
            content_img, style_img, device, config.optimization, precision=config.hardware.precision)

    output_path = runtime.setup_output_directory(config.output.output)
    content_name, style_name = Path(paths.content_path).stem, Path(paths.style_path).stem

    runner = optimization.OptimizationRunner(model, input_img, config, optimizer=optimizer,
                                             video_writer=video_writer, gif_collector=gif_collector)
    input_img, loss_metrics, elapsed = runner.run()
    for sink in (video_writer, gif_collector):
        if sink is not None:
            sink.close()

    runtime.save_outputs(input_img, loss_metrics, output_path, elapsed, SaveOptions(
        content_name=content_name, style_name=style_name, normalize=normalize,
        video_created=video_writer is not None, gif_created=gif_collector is not None,
        plot_losses=config.output.plot_losses))
    return input_img.detach().clamp(0, 1)


def style_transfer_batch(pairs: list[InputPaths], config, *, images_per_gpu: int | None = None) -> list[torch.Tensor]:
    """Several INDEPENDENT content/style pairs, one process per GPU (launch with torchrun).

    Not a tensor batch - ``gram_matrix`` folds a batch dimension into channels (reference
    core_model.py:56-57) - but N replicas of the single-image path: rank r runs pairs r, r + world, ...
    each with its own model targets and L-BFGS state, writes their PNGs, and one all-gather at the end
    hands every rank the full ordered list of result images (they must share one size).
    ``images_per_gpu`` of a rank's pairs run at the same time, each on its own stream (default 3:
    ``parallel.images_in_flight``) - same results, one image's optimizer update overlaps another's closure.
    """
    import copy  # noqa: PLC0415

    from . import parallel  # noqa: PLC0415

    _rank, local_rank, _world = parallel.init_distributed()
    if torch.cuda.is_available():              # "cuda" in the config then means this rank's GPU
        torch.cuda.set_device(local_rank % torch.cuda.device_count())

    def one(_index: int, paths: InputPaths) -> torch.Tensor:
        return style_transfer(paths, copy.deepcopy(config)).contiguous()
    return parallel.run_sharded(list(pairs), one, concurrent=images_per_gpu)
