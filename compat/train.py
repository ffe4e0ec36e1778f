"""``train.training`` of the absent gaussian-splatting submodule (/root/reference/src/gs/gs_training.py:7,46).

The optimisation loop (densification, Adam, checkpoints) is OUT OF SCOPE for this build (SURVEY.md section 8f row 4 covers
only what it would call: the differentiable rasterizer ``pgr_backward`` and ``distCUDA2``).  The name exists so that
``from train import training`` resolves; calling it says what is missing instead of failing somewhere inside."""


def training(*args, **kwargs):
    raise NotImplementedError(
        "train.training (the 3DGS optimisation loop) is not part of pegasus_amd: this build provides the render path, the "
        "differentiable rasterizer (diff_gaussian_rasterization, pgr_backward) and simple_knn.distCUDA2, not the trainer")
