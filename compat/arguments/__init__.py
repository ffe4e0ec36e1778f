"""Argument groups PEGASUS re-uses by appending to sys.argv (pegasus.py:152-154): only the fields the render
path reads are kept (sh_degree, source/model paths, white_background, data_device; the three pipeline flags)."""
from argparse import ArgumentParser, Namespace


class GroupParams:
    pass


class ParamGroup:
    def __init__(self, parser: ArgumentParser, name: str, fill_none=False):
        group = parser.add_argument_group(name)
        for key, value in vars(self).items():
            shorthand = key.startswith("_")
            key = key[1:] if shorthand else key
            t = type(value)
            value = value if not fill_none else None
            flags = ["--" + key] + (["-" + key[0:1]] if shorthand else [])
            if t == bool:
                group.add_argument(*flags, default=value, action="store_true")
            else:
                group.add_argument(*flags, default=value, type=t)

    def extract(self, args):
        group = GroupParams()
        for k, v in vars(args).items():
            if k in vars(self) or ("_" + k) in vars(self):
                setattr(group, k, v)
        return group


class ModelParams(ParamGroup):
    def __init__(self, parser, sentinel=False):
        self.sh_degree = 3
        self._source_path = ""
        self._model_path = ""
        self._images = "images"
        self._resolution = -1
        self._white_background = False
        self.data_device = "cuda"
        self.eval = False
        super().__init__(parser, "Loading Parameters", sentinel)


class PipelineParams(ParamGroup):
    def __init__(self, parser):
        self.convert_SHs_python = False
        self.compute_cov3D_python = False
        self.debug = False
        super().__init__(parser, "Pipeline Parameters")


class OptimizationParams(ParamGroup):
    def __init__(self, parser):
        self.iterations = 30_000
        super().__init__(parser, "Optimization Parameters")


def get_combined_args(parser: ArgumentParser):
    import sys
    return parser.parse_args(sys.argv[1:])
