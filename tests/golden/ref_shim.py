"""Import the reference's Python modules from /root/reference in the BUILD container only.

Used by make_golden.py to generate the fixtures in this directory; never imported by tests,
bench.py or the package (the reference does not exist on the GPU box).

The reference's modules import packages this image lacks (toml, cv2, c3d, anytree, tkinter,
PyQt5) and ask importlib.metadata for an installed 'pose2sim' distribution.  These are ordinary
ModuleNotFoundError / PackageNotFoundError conditions, handled by placing small
stand-in modules in sys.modules before the import:

* toml.load           -> tomli
* cv2.SVDecomp        -> numpy.linalg.svd (LAPACK; same mathematical object as OpenCV's Jacobi SVD).
                         A NaN input returns NaN factors (what a Jacobi sweep produces) instead of
                         LAPACK's LinAlgError.
* cv2.Rodrigues / getOptimalNewCameraMatrix / undistortPoints / projectPoints
                      -> pose2sim_amd.cvmath (restated from OpenCV's published algorithm: goldens of
                         the undistort path are self-consistent, not OpenCV-verified)
* anytree             -> a 40-line tree with the same pre-order RenderTree / PreOrderIter semantics
* c3d, tkinter, PyQt5, matplotlib Qt backend -> empty
"""
import importlib
import importlib.metadata
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'
sys.dont_write_bytecode = True
_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


class _Node:
    def __init__(self, name, parent=None, children=None, **kwargs):
        self.name = name
        self.children = []
        self.parent = None
        for k, v in kwargs.items():
            setattr(self, k, v)
        if parent is not None:
            parent.children.append(self)
            self.parent = parent
        for c in (children or []):
            c.parent = self
            self.children.append(c)


def _preorder(node):
    yield node
    for c in node.children:
        yield from _preorder(c)


class _RenderTree:
    def __init__(self, node):
        self.node = node

    def __iter__(self):
        for n in _preorder(self.node):
            yield ('', '', n)


class _DictImporter:
    def import_(self, data):
        data = dict(data)
        children = data.pop('children', [])
        name = data.pop('name')
        node = _Node(name, **data)
        for c in children:
            ch = self.import_(c)
            ch.parent = node
            node.children.append(ch)
        return node


def _cv2_module():
    from pose2sim_amd import cvmath
    cv2 = types.ModuleType('cv2')

    def SVDecomp(A):
        A = np.asarray(A, dtype=np.float64)
        if not np.isfinite(A).all():
            m, n = A.shape
            return np.full((n, 1), np.nan), np.full((m, n), np.nan), np.full((n, n), np.nan)
        U, S, Vt = np.linalg.svd(A, full_matrices=False)
        return S.reshape(-1, 1), U, Vt

    def Rodrigues(r):
        r = np.asarray(r, dtype=np.float64)
        if r.size == 9:                                  # matrix -> vector (calibration converters)
            return cvmath.rodrigues_from_matrix(r).reshape(3, 1), None
        return cvmath.rodrigues(r), None

    def getOptimalNewCameraMatrix(K, dist, size, alpha, new_size=None):
        return cvmath.get_optimal_new_camera_matrix(K, dist, size, alpha, new_size), (0, 0, 0, 0)

    def undistortPoints(points, K, dist, R, P):
        pts = np.asarray(points)
        out = cvmath.undistort_points(pts.reshape(-1, 2), K, dist, P)
        return out.astype(np.float32).reshape(-1, 1, 2)

    def projectPoints(Q, R, T, K, dist):
        uv = cvmath.project_points(np.asarray(Q, dtype=np.float64).reshape(-1, 3), R, T, K, dist)
        return uv.reshape(-1, 1, 2), None

    class VideoCapture:
        def __init__(self, *a, **k):
            pass

        def read(self):
            return False, None

        def get(self, prop):
            return 0.0

    cv2.SVDecomp = SVDecomp
    cv2.Rodrigues = Rodrigues
    cv2.getOptimalNewCameraMatrix = getOptimalNewCameraMatrix
    cv2.undistortPoints = undistortPoints
    cv2.projectPoints = projectPoints
    cv2.VideoCapture = VideoCapture
    cv2.CAP_PROP_FPS = 5
    return cv2


def install():
    """Install the stand-ins; idempotent."""
    if 'Pose2Sim' in sys.modules:
        return
    import tomli
    toml = types.ModuleType('toml')

    def _load(path):
        with open(path, 'rb') as f:
            return tomli.load(f)
    toml.load = _load
    sys.modules['toml'] = toml

    sys.modules['cv2'] = _cv2_module()
    sys.modules['c3d'] = types.ModuleType('c3d')
    sys.modules['tkinter'] = types.ModuleType('tkinter')

    anytree = types.ModuleType('anytree')
    anytree.Node = _Node
    anytree.RenderTree = _RenderTree
    anytree.PreOrderIter = _preorder
    importer = types.ModuleType('anytree.importer')
    importer.DictImporter = _DictImporter
    anytree.importer = importer
    sys.modules['anytree'] = anytree
    sys.modules['anytree.importer'] = importer

    pyqt = types.ModuleType('PyQt5')
    qtw = types.ModuleType('PyQt5.QtWidgets')
    for n in ('QMainWindow', 'QApplication', 'QWidget', 'QTabWidget', 'QVBoxLayout'):
        setattr(qtw, n, type(n, (), {}))
    pyqt.QtWidgets = qtw
    sys.modules['PyQt5'] = pyqt
    sys.modules['PyQt5.QtWidgets'] = qtw
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.backends  # noqa: F401
    qt5agg = types.ModuleType('matplotlib.backends.backend_qt5agg')
    qt5agg.FigureCanvasQTAgg = type('FigureCanvasQTAgg', (), {})
    qt5agg.NavigationToolbar2QT = type('NavigationToolbar2QT', (), {})
    sys.modules['matplotlib.backends.backend_qt5agg'] = qt5agg

    _orig_version = importlib.metadata.version

    def _version(name):
        if name.lower() == 'pose2sim':
            return '0.0.0-reference'
        return _orig_version(name)
    importlib.metadata.version = _version

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load():
    """Returns (common, triangulation, personAssociation, skeletons) reference modules."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('the reference is only present in the build container')
    install()
    common = importlib.import_module('Pose2Sim.common')
    tri = importlib.import_module('Pose2Sim.triangulation')
    pa = importlib.import_module('Pose2Sim.personAssociation')
    sk = importlib.import_module('Pose2Sim.skeletons')
    return common, tri, pa, sk
