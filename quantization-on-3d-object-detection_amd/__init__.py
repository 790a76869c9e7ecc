"""amdq3d — MI355X-native quantized CenterPoint inference path (Python host side).

The directory name follows the project naming scheme and is not a valid Python identifier; import the
package through the top-level alias module `amdq3d` (repo root), which points here.
"""
__version__ = "0.1.0"
