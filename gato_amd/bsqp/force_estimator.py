"""Force-hypothesis generator of the batched MPC loop: the host-side mirror of the reference's `examples/force_estimator.py:4-155`
(`ForceEstimator`), numpy only.  Hypothesis 0 is the smoothed estimate, 1 is "no force", 2 extrapolates the estimate along its
momentum, the remaining B-3 sit on a Fibonacci sphere of radius `radius` (randomly re-oriented after every update) around a blend
of the smoothed and raw estimates; `update()` pulls the estimate towards the hypothesis that predicted the measured state best and
adapts the radius.  Same attribute names, same arithmetic (float32 state, numpy's global random stream for the re-orientation), so
`tests/test_force_estimator.py` reproduces a fixture generated from the imported reference bit for bit."""
import numpy as np


class ForceEstimator:
    def __init__(self, batch_size, initial_radius=10.0, min_radius=1.0, max_radius=100.0, smoothing_factor=0.3):
        assert batch_size > 3, "Batch size must be > 3 for exploitation + exploration strategy"
        self.batch_size = batch_size
        self.dim = 6
        self.radius = initial_radius
        self.min_radius, self.max_radius = min_radius, max_radius
        self.radius_increase_factor, self.radius_decrease_factor = 1.05, 0.95
        self.smoothing_factor = smoothing_factor
        self.sphere_dirs = self._fibonacci_sphere(batch_size - 3)
        self._zero_state()

    def _zero_state(self):
        self.estimate = np.zeros(self.dim, dtype=np.float32)
        self.momentum = np.zeros(self.dim, dtype=np.float32)
        self.smoothed_estimate = np.zeros(self.dim, dtype=np.float32)
        self.confidence = 0.0
        self.error_history = []
        self.current_rotation = np.eye(3, dtype=np.float32)

    @staticmethod
    def _fibonacci_sphere(n):
        """n unit vectors along a Fibonacci spiral, y from +1 down to -1 (force_estimator.py:30-59)"""
        pts = np.zeros((n, 3), dtype=np.float32)
        if n == 0:
            return pts
        golden = (1 + np.sqrt(5)) / 2
        for i in range(n):
            y = 1 - (2 * i / (n - 1)) if n > 1 else 0
            ring = np.sqrt(1 - y * y)
            ang = 2 * np.pi * i / golden
            pts[i] = (ring * np.cos(ang), y, ring * np.sin(ang))
        return pts

    @staticmethod
    def _random_rotation_matrix():
        """uniformly random rotation from a uniformly random unit quaternion (three draws of numpy's global stream)"""
        u1, u2, u3 = np.random.rand(3)
        x = np.sqrt(1.0 - u1) * np.sin(2.0 * np.pi * u2)
        y = np.sqrt(1.0 - u1) * np.cos(2.0 * np.pi * u2)
        z = np.sqrt(u1) * np.sin(2.0 * np.pi * u3)
        w = np.sqrt(u1) * np.cos(2.0 * np.pi * u3)
        xx, yy, zz, xy, xz, yz, wx, wy, wz = x * x, y * y, z * z, x * y, x * z, y * z, w * x, w * y, w * z
        return np.array([[1.0 - 2.0 * (yy + zz), 2.0 * (xy - wz), 2.0 * (xz + wy)],
                         [2.0 * (xy + wz), 1.0 - 2.0 * (xx + zz), 2.0 * (yz - wx)],
                         [2.0 * (xz - wy), 2.0 * (yz + wx), 1.0 - 2.0 * (xx + yy)]], dtype=np.float32)

    def generate_batch(self):
        batch = np.zeros((self.batch_size, 6), dtype=np.float32)
        batch[0] = self.smoothed_estimate
        batch[2] = self.smoothed_estimate + 0.5 * self.momentum
        centre = 0.7 * self.smoothed_estimate[:3] + 0.3 * self.estimate[:3]
        for i in range(3, self.batch_size):
            batch[i, :3] = centre + self.radius * (self.current_rotation @ self.sphere_dirs[i - 3])
            batch[i, 3:] = self.smoothed_estimate[3:]
        return batch

    def update(self, best_idx, prediction_errors, alpha=0.5, beta=0.8):
        self.error_history.append(np.min(prediction_errors))
        best_force = self.generate_batch()[best_idx, :]
        self.momentum = beta * self.momentum + (1 - beta) * (best_force - self.estimate)
        raw_update = alpha * best_force + (1 - alpha) * self.estimate
        self.estimate = 0.8 * self.estimate + 0.2 * (raw_update + 0.5 * self.momentum)
        self.smoothed_estimate = (1 - self.smoothing_factor) * self.smoothed_estimate + self.smoothing_factor * self.estimate
        if best_idx < 3:   # an exploiting hypothesis won: shrink the exploration sphere
            self.radius *= self.radius_decrease_factor
            self.confidence = min(1.0, self.confidence + 0.05)
        else:
            self.radius *= self.radius_increase_factor
            self.confidence = max(0.0, self.confidence - 0.1)
        self.radius = np.clip(self.radius, self.min_radius, self.max_radius)
        if len(self.error_history) > 5:
            recent = self.error_history[-5:]
            if np.std(recent) < 0.01:
                self.radius *= 0.9
            elif recent[-1] > 1.5 * np.mean(recent[:-1]):
                self.radius *= 1.3
                self.confidence *= 0.5
            self.radius = np.clip(self.radius, self.min_radius, self.max_radius)
        self.current_rotation = self._random_rotation_matrix()

    def reset(self):
        self._zero_state()
        self.radius = 10.0

    def get_stats(self):
        return {"current_estimate": self.estimate.copy(), "smoothed_estimate": self.smoothed_estimate.copy(), "momentum": self.momentum.copy(),
                "radius": self.radius, "confidence": self.confidence, "recent_error": self.error_history[-1] if self.error_history else np.inf}
