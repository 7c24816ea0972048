"""Synthetic tracker scene for end-to-end measurements and tests: moving pedestrian-shaped boxes over synthetic
frames, crops cut by `BUSCA.get_image_crops` (so track memories are device-resident), lost tracks with Kalman
candidates - the caller contract of byte_tracker.py:367-397 / deep_sort tracker.py:129-189 without a real tracker."""
import time

import numpy as np
import torch

from . import synth


class SimTrack:
    """Track protocol of associate_embeddings: images_mem, tlwh_mem, scale, tlwh (+ tlbr for center_distance)."""

    def __init__(self, tlwh, image, scale=1.0):
        self.tlwh_mem = [np.asarray(tlwh, dtype=np.float64)]
        self.images_mem = [image]
        self.scale = scale

    @property
    def tlwh(self):
        return self.tlwh_mem[-1]

    @property
    def tlbr(self):
        r = self.tlwh_mem[-1].copy()
        r[2:] += r[:2]
        return r

    def update(self, tlwh, image):
        self.tlwh_mem.append(np.asarray(tlwh, dtype=np.float64))
        self.images_mem.append(image)


class SimScene:
    def __init__(self, model, n_objects=150, H=1080, W=1920, seed=7):
        self.model, self.H, self.W, self.n = model, H, W, n_objects
        self.frame_id = 0
        self.seed = seed
        h = synth.uniform(seed, "sh", (n_objects,), 80, 320).astype(np.float64)
        w = h * synth.uniform(seed, "sar", (n_objects,), 0.3, 0.5)
        x = synth.uniform(seed, "sx", (n_objects,), 0, W - 100).astype(np.float64)
        y = synth.uniform(seed, "sy", (n_objects,), 0, H - 200).astype(np.float64)
        self.box = np.stack([x, y, w, h], 1)
        self.vel = np.stack([synth.uniform(seed, "vx", (n_objects,), -4, 4), synth.uniform(seed, "vy", (n_objects,), -1, 1)], 1).astype(np.float64)
        self.base = synth.randint_u8(seed, "frame", (H, W, 3))
        self.tracks = []

    def next_frame(self):
        """Advance one frame: returns (frame u8 BGR, tlwh boxes [n,4])."""
        self.frame_id += 1
        self.box[:, :2] += self.vel
        frame = np.roll(self.base, 7 * self.frame_id, axis=1)
        return frame, self.box.copy()

    def warm_up(self, n_frames):
        """Run n frames where every object is detected: builds the per-track memories (device-resident crops)."""
        for _ in range(n_frames):
            frame, boxes = self.next_frame()
            tlbr = boxes.copy()
            tlbr[:, 2:] += tlbr[:, :2]
            crops = self.model.get_image_crops(frame, tlbr, normalize=False)
            if not self.tracks:
                self.tracks = [SimTrack(boxes[i], crops[i]) for i in range(self.n)]
            else:
                for i, t in enumerate(self.tracks):
                    t.update(boxes[i], crops[i])

    def step_inputs(self, n_lost):
        """One frame in which the first n_lost objects lose their detection: returns (lost tracks, detection objects,
        Kalman candidates) exactly as a tracker hands them to associate_embeddings."""
        return self.crop_inputs(*self.next_frame(), n_lost)

    def crop_inputs(self, frame, boxes, n_lost, per_detection=False):
        """The tracker-side half of step_inputs for a frame already produced (next_frame): the two get_image_crops calls of an update - detections,
        Kalman boxes of the lost tracks - and the objects built from them."""
        tlbr = boxes.copy()
        tlbr[:, 2:] += tlbr[:, :2]
        det_idx = np.arange(n_lost, self.n)
        t0 = time.perf_counter()
        if per_detection:
            # the call pattern of the UNCHANGED StrongSORT / GHOST adapters (deep_sort/tracker.py:126,273,291): one get_image_crops per detection / per
            # Kalman box, no frame scope - every call reads the live host frame
            det_crops = [self.model.get_image_crops(frame, tlbr[i:i + 1], normalize=False)[0] for i in det_idx]
            kal_crops = [self.model.get_image_crops(frame, tlbr[i:i + 1], normalize=False)[0] for i in range(n_lost)]
        else:
            with self.model.frame(frame):            # one upload for both calls (explicit scope: an integrated adapter's `with model.frame(img):` around update)
                det_crops = self.model.get_image_crops(frame, tlbr[det_idx], normalize=False)
                kal_crops = self.model.get_image_crops(frame, tlbr[:n_lost], normalize=False)
        self.last_crop_call_count = (len(det_idx) + n_lost) if per_detection else 2
        # the crops are in HBM when the tracker's stream is done; their host copy (lazy mode: a side stream into pinned memory) keeps flowing
        # under whatever follows and is only waited for by a host read of the pixels
        torch.cuda.current_stream().synchronize()
        self.last_crop_calls_s = time.perf_counter() - t0            # the library's share of this method (the rest is this simulator's bookkeeping)
        dets = [SimTrack(boxes[i], det_crops[k]) for k, i in enumerate(det_idx)]
        lost = self.tracks[:n_lost]
        kalman = [SimTrack(boxes[i], kal_crops[i]) for i in range(n_lost)]
        return lost, dets, kalman
