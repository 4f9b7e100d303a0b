"""Packaging for the MI355X `woltka classify` hot path.

    python setup.py build_ext --inplace     # hipcc -> woltka_amd/libwoltka_hip.so
    pip install -e .                         # + the `woltka-amd` command

The shared library is built in-tree by the same recipe as
``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
"""
import os
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildNative(Command):
    description = 'compile the HIP kernels and the C ABI for gfx950'
    user_options = [('inplace', 'i', 'accepted for familiarity; always in-tree')]

    def initialize_options(self):
        self.inplace = True

    def finalize_options(self):
        pass

    def run(self):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build_native()


class BuildPy(build_py):
    def run(self):
        self.run_command('build_ext')
        super().run()


setup(
    name='woltka-amd',
    version='0.1.7',
    description='MI355X-native classify path of Woltka (HIP kernels behind a C ABI)',
    packages=find_packages(include=['woltka_amd', 'woltka_amd.*']),
    package_data={'woltka_amd': ['libwoltka_hip.so', 'csrc/*']},
    python_requires='>=3.8',
    install_requires=['numpy', 'click'],
    entry_points={'console_scripts': ['woltka-amd=woltka_amd.cli:cli']},
    cmdclass={'build_ext': BuildNative, 'build_py': BuildPy},
)
