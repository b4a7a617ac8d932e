"""`import pixelssl` -> the MI355X engine (pixelssl_amd), so that plugins written against the reference package
(`pixelssl.criterion_template.TaskCriterion`, `pixelssl.nn.func`, `pixelssl.utils.logger`, `from pixelssl.ssl_algorithm
import ssl_base`, ...) import unchanged.  Every pixelssl_amd submodule is registered under the pixelssl.* name as the SAME
module object (no second copy of any class).  Put this repository on sys.path instead of the reference checkout."""
import importlib
import pkgutil
import sys

import pixelssl_amd as _impl


def _alias():
    for m in pkgutil.walk_packages(_impl.__path__, "pixelssl_amd."):
        if m.name.split(".")[1] in ("csrc", "build", "libpixelhip"):      # native sources / the C-ABI library itself
            continue
        importlib.import_module(m.name)
    for name, mod in list(sys.modules.items()):
        if name == "pixelssl_amd" or name.startswith("pixelssl_amd."):
            sys.modules["pixelssl" + name[len("pixelssl_amd"):]] = mod


_alias()
