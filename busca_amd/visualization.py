"""Debug drawing is out of the hot path (busca/visualization.py is only reached with
--online-visualization).  The adapters import `plot_box` unconditionally (byte_tracker.py:21), so the
name exists; it needs OpenCV at call time."""


def plot_box(*args, **kwargs):
    try:
        import cv2  # noqa: F401
    except ImportError as e:
        raise RuntimeError("busca.visualization.plot_box needs OpenCV, which is not part of busca_amd") from e
    raise NotImplementedError("online visualisation is outside the MI355X hot path; use the reference's busca/visualization.py")


def create_batch_image(*args, **kwargs):
    raise NotImplementedError("online visualisation is outside the MI355X hot path")
