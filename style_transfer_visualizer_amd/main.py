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


def style_transfer(paths: InputPaths, config, *, video_writer=None, gif_collector=None, output_tag: str = "",
                   tag_png: bool = True) -> torch.Tensor:
    """Run one style transfer; returns the optimised image clamped to [0, 1].

    ``output_tag`` (``style_transfer_batch``): appended to the names of this run's loss CSV and loss plot - and, with
    ``tag_png``, of its PNG - so that several runs writing into one directory do not overwrite each other."""
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
    content_name, style_name = Path(paths.content_path).stem, Path(paths.style_path).stem + (output_tag if tag_png else "")
    if output_tag and config.output.log_loss:
        log_path = Path(config.output.log_loss)
        config.output.log_loss = str(log_path.with_name(log_path.stem + output_tag + log_path.suffix))

    runner = optimization.OptimizationRunner(model, input_img, config, optimizer=optimizer,
                                             video_writer=video_writer, gif_collector=gif_collector)
    input_img, loss_metrics, elapsed = runner.run()
    for sink in (video_writer, gif_collector):
        if sink is not None:
            sink.close()

    runtime.save_outputs(input_img, loss_metrics, output_path, elapsed, SaveOptions(
        content_name=content_name, style_name=style_name, normalize=normalize,
        video_created=video_writer is not None, gif_created=gif_collector is not None,
        plot_losses=config.output.plot_losses),
        **({"plot_name": f"loss_plot{output_tag}.png"} if output_tag else {}))      # (untagged: the reference's call, runtime/output.py:55)
    return input_img.detach().clamp(0, 1)


def style_transfer_batch(pairs: list[InputPaths], config, *, images_per_gpu: int | None = None) -> list[torch.Tensor]:
    """Several INDEPENDENT content/style pairs, one process per GPU (launch with torchrun).

    Not a tensor batch - ``gram_matrix`` folds a batch dimension into channels (reference
    core_model.py:56-57) - but N replicas of the single-image path: rank r runs pairs r, r + world, ...
    each with its own model targets and L-BFGS state, writes their PNGs, and one all-gather at the end
    hands every rank the full ordered list of result images (they must share one size).
    ``images_per_gpu`` of a rank's pairs run at the same time, each on its own stream (opt-in: default 1, or
    ``STV_IMAGES_PER_GPU``; ``parallel.images_in_flight``) - same results, one image's optimizer update overlaps
    another's closure, at that many times the device memory.  Every pair works on its own copy of ``config``; with
    more than one pair the loss CSV (``output.log_loss``) and the loss plot carry the pair's index, and so does the PNG
    of a pair whose content/style names occur more than once in ``pairs`` - no two pairs write one file.
    """
    import copy  # noqa: PLC0415

    from . import parallel  # noqa: PLC0415

    _rank, local_rank, _world = parallel.init_distributed()
    if torch.cuda.is_available():              # "cuda" in the config then means this rank's GPU
        torch.cuda.set_device(local_rank % torch.cuda.device_count())

    stems = [(Path(p.content_path).stem, Path(p.style_path).stem) for p in pairs]

    def one(index: int, paths: InputPaths) -> torch.Tensor:
        tag = f"_{index:03d}" if len(pairs) > 1 else ""
        # (a pair whose names are unique in the batch keeps the reference's PNG name, runtime/output.py:46-52)
        return style_transfer(paths, copy.deepcopy(config), output_tag=tag, tag_png=stems.count(stems[index]) > 1).contiguous()
    return parallel.run_sharded(list(pairs), one, concurrent=images_per_gpu)
