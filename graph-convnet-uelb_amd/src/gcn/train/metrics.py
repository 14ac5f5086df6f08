"""Per-epoch bookkeeping of a training run (reference: src/gcn/train/metrics.py:10-367, GCNMetricsLogger, and the
base class it extends).  Same recording calls, with the argument meaning the reference's trainer uses
(src/gcn/train/trainer.py:514, 541, 553, 571), the same ``*_list`` attributes and the same keys from
``get_final_metrics``.  Results are written as JSON and a text summary; the reference also pickles the lists and draws
matplotlib figures, which are not reproduced here."""
import json
import os
import time


class GCNMetricsLogger:
    RL_KEYS = ("reward", "advantage", "entropy", "load_factor", "baseline", "complete_paths_rate", "finite_solution_rate",
               "avg_finite_load_factor", "avg_path_length", "commodity_success_rate", "capacity_violation_rate")
    RL_CSV_COLUMNS = ("reward", "reward_std", "advantage", "advantage_std", "entropy", "load_factor", "load_factor_std", "baseline",
                      "complete_paths_rate", "finite_solution_rate", "avg_finite_load_factor", "avg_path_length",
                      "commodity_success_rate", "capacity_violation_rate")

    def __init__(self, save_dir="logs"):
        self.save_dir = save_dir
        os.makedirs(save_dir, exist_ok=True)
        self.timestamp = time.strftime("%Y%m%d_%H%M%S")
        self.train_loss_list, self.train_err_edges_list, self.train_time_list = [], [], []
        self.val_approximation_rate_list, self.val_time_list, self.val_epochs = [], [], []
        self.test_approximation_rate_list, self.test_time_list, self.test_epochs = [], [], []
        self.val_comp_rate_list, self.val_comp_sample_rate_list = [], []
        self.test_comp_rate_list, self.test_comp_sample_rate_list = [], []
        self.total_train_time = self.total_val_time = self.total_test_time = 0.0
        for k in self.RL_KEYS:
            setattr(self, f"rl_{k}_list", [])

    # ---- recording (call shapes of the reference)
    def log_train_metrics(self, loss, edge_error, train_time=None):
        self.train_loss_list.append(float(loss))
        self.train_err_edges_list.append(float(edge_error))
        if train_time is not None:
            self.train_time_list.append(float(train_time))
            self.total_train_time += float(train_time)

    def _log_eval(self, which, approximation_rate, seconds, epoch, comp_rate, comp_sample_rate):
        getattr(self, f"{which}_approximation_rate_list").append(float(approximation_rate))
        if seconds is not None:
            getattr(self, f"{which}_time_list").append(float(seconds))
            setattr(self, f"total_{which}_time", getattr(self, f"total_{which}_time") + float(seconds))
        if epoch is not None:
            getattr(self, f"{which}_epochs").append(int(epoch))
        if comp_rate is not None:
            getattr(self, f"{which}_comp_rate_list").append(float(comp_rate))
        if comp_sample_rate is not None:
            getattr(self, f"{which}_comp_sample_rate_list").append(float(comp_sample_rate))

    def log_val_metrics(self, approximation_rate, val_time=None, epoch=None, comp_rate=None, comp_sample_rate=None):
        self._log_eval("val", approximation_rate, val_time, epoch, comp_rate, comp_sample_rate)

    def log_test_metrics(self, approximation_rate, test_time=None, epoch=None, comp_rate=None, comp_sample_rate=None):
        self._log_eval("test", approximation_rate, test_time, epoch, comp_rate, comp_sample_rate)

    def log_rl_metrics(self, epoch, rl_metrics):
        """One row per epoch, in memory and appended to rl_gcn_metrics_<timestamp>.csv (the reference's columns)."""
        for k in self.RL_KEYS:
            getattr(self, f"rl_{k}_list").append(float(rl_metrics.get(k, 0.0)))
        path = os.path.join(self.save_dir, f"rl_gcn_metrics_{self.timestamp}.csv")
        new = not os.path.exists(path)
        with open(path, "a", encoding="utf-8") as f:
            if new:
                f.write("epoch," + ",".join(self.RL_CSV_COLUMNS) + "\n")
            f.write(f"{int(epoch)}," + ",".join(f"{float(rl_metrics.get(c, 0.0)):.6f}" for c in self.RL_CSV_COLUMNS) + "\n")

    # ---- reporting
    def get_final_metrics(self):
        last = lambda xs: xs[-1] if xs else 0.0                      # noqa: E731
        mean = lambda xs: sum(xs) / len(xs) if xs else 0.0           # noqa: E731
        return {
            "final_train_loss": last(self.train_loss_list),
            "final_edge_error": last(self.train_err_edges_list),
            "final_val_approximation_rate": last(self.val_approximation_rate_list),
            "final_test_approximation_rate": last(self.test_approximation_rate_list),
            "best_val_approximation_rate": max(self.val_approximation_rate_list, default=0.0),
            "best_test_approximation_rate": max(self.test_approximation_rate_list, default=0.0),
            "total_train_time": self.total_train_time,
            "total_test_time": self.total_test_time,
            "avg_train_time_per_epoch": mean(self.train_time_list),
            "avg_test_time_per_epoch": mean(self.test_time_list),
        }

    def calculate_time_per_data(self, num_train_data, num_test_data):
        ntrain = num_train_data * len(self.train_time_list)
        ntest = num_test_data * len(self.test_time_list)
        return {"train_time_per_data": self.total_train_time / ntrain if ntrain else 0.0,
                "test_time_per_data": self.total_test_time / ntest if ntest else 0.0,
                "total_train_samples": ntrain, "total_test_samples": ntest}

    def _lists(self):
        return {k: v for k, v in vars(self).items() if k.endswith("_list") or k.endswith("_epochs")}

    def save_results(self, config_info=None):
        """Writes <save_dir>/training_results_<timestamp>.json (all lists, final metrics, config_info) and a .txt summary;
        returns both paths."""
        info = dict(config_info or {})
        per = self.calculate_time_per_data(info.get("num_train_data", 0), info.get("num_test_data", 0))
        results = {"timestamp": self.timestamp, **self._lists(), "final_metrics": self.get_final_metrics(),
                   "time_per_data": per, "config_info": info}
        jpath = os.path.join(self.save_dir, f"training_results_{self.timestamp}.json")
        with open(jpath, "w", encoding="utf-8") as f:
            json.dump(results, f, indent=1, default=str)
        tpath = os.path.join(self.save_dir, f"training_results_{self.timestamp}.txt")
        with open(tpath, "w", encoding="utf-8") as f:
            for k, v in results["final_metrics"].items():
                f.write(f"{k}: {v}\n")
            for k, v in per.items():
                f.write(f"{k}: {v}\n")
        return jpath, tpath

    def print_summary(self, config_info=None):
        m = self.get_final_metrics()
        print("\nTRAINING SUMMARY")
        print(f"final train loss {m['final_train_loss']:.4f}   final train edge error {m['final_edge_error']:.2f}%")
        if self.val_approximation_rate_list:
            print(f"final val approximation rate {m['final_val_approximation_rate']:.2f}%  (best {m['best_val_approximation_rate']:.2f}%)")
        if self.test_approximation_rate_list:
            print(f"final test approximation rate {m['final_test_approximation_rate']:.2f}%  (best {m['best_test_approximation_rate']:.2f}%)")
        info = config_info or {}
        per = self.calculate_time_per_data(info.get("num_train_data", 0), info.get("num_test_data", 0))
        print(f"training time {m['total_train_time']:.2f}s ({m['avg_train_time_per_epoch']:.2f}s per epoch, "
              f"{per['train_time_per_data']:.4f}s per sample); test time {m['total_test_time']:.2f}s", flush=True)

    def print_evaluation_summary(self, config_info=None):
        print("\nEVALUATION SUMMARY")
        if self.val_approximation_rate_list:
            print(f"validation: approximation rate {self.val_approximation_rate_list[-1]:.2f}%  "
                  f"time {(self.val_time_list[-1] if self.val_time_list else 0.0):.2f}s")
        if self.test_approximation_rate_list:
            print(f"test: approximation rate {self.test_approximation_rate_list[-1]:.2f}%  "
                  f"time {(self.test_time_list[-1] if self.test_time_list else 0.0):.2f}s")
        if config_info:
            print(f"config {config_info.get('config_file', '?')}  model hash {config_info.get('model_hash', '?')}", flush=True)


def metrics_to_str(epoch, time, learning_rate, loss, mean_maximum_load_factor, gt_load_factor, approximation_rate,
                   infeasible_rate, comp_rate=None, comp_sample_rate=None):
    """One line per evaluation, the reference's fields in the reference's order (metrics.py:369-386)."""
    s = (f"epoch:{epoch:02d}\texecution time:{time:.8f}s\tlr:{learning_rate:.2e}\tloss:{loss:.4f}\t"
         f"mean_maximum_load_factor:{mean_maximum_load_factor:.3f}\tgt_load_factor:{gt_load_factor:.3f}\t"
         f"approximation_rate:{approximation_rate:.3f}\tinfeasible_rate:{infeasible_rate:.3f}")
    if comp_rate is not None:
        s += f"\tcomp_rate:{comp_rate:.2f}%"
    if comp_sample_rate is not None:
        s += f"\tcomp_sample_rate:{comp_sample_rate:.2f}%"
    return s
