"""Import aliases for scripts written against the reference's package name.

`sdirt_amd/compat/` on PYTHONPATH (or `sdirt_amd.compat.install()`) makes `import deeplens`,
`from deeplens.psfnet import PSFNet`, `from deeplens.utils import set_seed, set_logger` ... resolve to this
package: each `deeplens.<module>` is a few re-export lines over the `sdirt_amd` module of the same role.
"""
import os
import sys


def path():
    """The directory to put on sys.path / PYTHONPATH."""
    return os.path.dirname(os.path.abspath(__file__))


def install():
    """Make `import deeplens` resolve to the aliases for this process."""
    if path() not in sys.path:
        sys.path.insert(0, path())
