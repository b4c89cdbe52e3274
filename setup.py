"""Build hook of the installable artefact: `pip install .` compiles the native libraries and ships them -- with the C-ABI
headers -- INSIDE the package, so that the installed tree works with the source tree gone.

The reference ships a wheel with its native module inside (Cargo.toml:16,113-117: an abi3 cdylib built by maturin;
pyproject.toml:58 the console entry).  Here the native side is `make -C deepchopper_amd/csrc`:
libdeepchopper_hip.so (hipcc, --offload-arch=gfx950; cross-compiles without a GPU) and libdeepchopper_io.so (g++ + zlib),
placed in deepchopper_amd/lib/, and include/*.h copied to deepchopper_amd/include/ (`_lib.HEADER_PATH` / `_iolib.HEADER_PATH`
look there first).  Metadata, entry points and package data are declared in pyproject.toml."""
import glob
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildNative(build_py):
    def run(self):
        csrc = os.path.join(ROOT, "deepchopper_amd", "csrc")
        jobs = os.environ.get("DC_BUILD_JOBS", str(min(8, os.cpu_count() or 1)))
        subprocess.run(["make", "-C", csrc, f"-j{jobs}"], check=True)
        inc = os.path.join(ROOT, "deepchopper_amd", "include")
        os.makedirs(inc, exist_ok=True)
        for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
            shutil.copy2(h, inc)
        super().run()


def legacy_metadata() -> dict:
    """setuptools < 61 (this image ships 59.6) does not read pyproject.toml's [project] table: hand it the same metadata."""
    import setuptools

    if int(setuptools.__version__.split(".")[0]) >= 61:
        return {}
    try:
        import tomllib as toml  # Python >= 3.11
    except ImportError:
        import tomli as toml
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
        cfg = toml.load(fh)
    proj, st = cfg["project"], cfg["tool"]["setuptools"]
    return {"name": proj["name"], "version": proj["version"], "description": proj["description"],
            "python_requires": proj["requires-python"], "install_requires": proj["dependencies"],
            "packages": st["packages"], "package_data": st["package-data"],
            "entry_points": {"console_scripts": [f"{k} = {v}" for k, v in proj["scripts"].items()]}}


setup(cmdclass={"build_py": BuildNative}, **legacy_metadata())
