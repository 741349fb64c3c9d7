"""Import the product package (its directory name has a hyphen, so it is loaded by path)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    name = "h264_lab_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "h264-lab_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# H264E_EMU_LIB_OVERRIDE: run the emulation tests against another build of the same sources (tools/sanitize_cpu.sh: ASan + UBSan)
EMU_LIB = os.environ.get("H264E_EMU_LIB_OVERRIDE") or os.path.join(ROOT, "tests", "emu", "build", "libh264e_emu.so")
EMU_REV_LIB = os.path.join(ROOT, "tests", "emu", "build", "libh264e_emu_rev.so")
