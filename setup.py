"""Installable form of the drop-in (the counterpart of the reference's linemodLevelup/setup.py:19-58, which builds the
pybind11 module with CMake): `pip install .` / `python setup.py build_ext --inplace` compiles 6dpose_amd/libamdlinemod.so with
hipcc for gfx950 (6dpose_amd/csrc/Makefile) and installs the module under the reference's own name,

    import linemodLevelup_pybind

together with the shared library next to it.  The in-tree use (PYTHONPATH=6dpose_amd, what the tests and bench.py do) needs no
installation."""
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "6dpose_amd")


class BuildWithHip(build_py):
    def run(self):
        env = dict(os.environ)
        env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], env=env)
        super().run()
        for name in ("libamdlinemod.so",):                      # the library travels with the module that loads it
            shutil.copy2(os.path.join(PKG, name), os.path.join(self.build_lib, name))


setup(
    name="linemodLevelup-amd",
    version="0.2.0",
    description="MI355X-native linemodLevelup Detector / poseRefine (drop-in for linemodLevelup_pybind)",
    package_dir={"": "6dpose_amd"},
    py_modules=["linemodLevelup_pybind", "sharded", "views"],
    cmdclass={"build_py": BuildWithHip},
    python_requires=">=3.8",
    install_requires=["numpy"],
)
