"""get_model with the reference's dispatch (itr/modalmodule/__init__.py:4-19).  The reference's config hook
renames 'VSE++' to 'VSE_PP' while get_model only matches 'VSE++' (KeyError); both spell here."""
from . import Models, ImgEncoder, TextEncoder, Objectives, Fusionmodule, utils  # noqa: F401

_BUILT = {'VSE++': 'VSE_PP', 'VSE_PP': 'VSE_PP', 'SCAN': 'SCAN', 'SGRAF': 'SGRAF', 'SAEM': 'SAEM', 'CAMERA': 'CAMERA',
          'VSRN': 'VSRN'}


def get_model(config):
    name = config['name']
    if name not in _BUILT:
        raise KeyError(f'No model is named {config["name"]}')
    cls = getattr(Models, _BUILT[name], None)
    if cls is None:
        raise NotImplementedError("%s is not built yet in this round (see DESIGN.md, coverage table)" % name)
    return cls(config)
