"""DataLoader.next(cl_ny, type) with the reference's restart-and-shuffle semantics
(Speech_enhancement_by_AAS/data_loader.py:8-83)."""
from .loader_functions import FeatDataset, FeatLoader, FeatLoader_paired, FeatSampler


class DataLoader():
    def __init__(self, batch_size, paired=False, tr_cl_manifest="", tr_ny_manifest="", trsub_manifest="",
                 val_manifest="", val2_manifest="", labels=None, num_workers=1):
        self.batch_size, self.labels, self.num_workers = batch_size, labels, num_workers
        self.Loader = FeatLoader_paired if paired else FeatLoader
        self._ds, self._sp, self._it = {}, {}, {}
        for key, manifest, sampled in (("cl/train", tr_cl_manifest, True), ("ny/train", tr_ny_manifest, True),
                                       ("ny/trsub", trsub_manifest, False), ("ny/val", val_manifest, False),
                                       ("ny/val2", val2_manifest, False)):
            if len(manifest) > 0:
                self._ds[key] = FeatDataset(manifest=manifest, labels=labels)
                if sampled:
                    self._sp[key] = FeatSampler(self._ds[key], batch_size=batch_size)
                self._it[key] = self._make(key)

    def _make(self, key):
        if key in self._sp:
            return iter(self.Loader(self._ds[key], num_workers=self.num_workers, batch_sampler=self._sp[key]))
        return iter(self.Loader(self._ds[key], batch_size=self.batch_size, num_workers=self.num_workers))

    def num_batches(self, type):
        ds = self._ds["ny/" + type]
        return (len(ds) + self.batch_size - 1) // self.batch_size

    def next(self, cl_ny="", type=""):
        key = "%s/%s" % (cl_ny, type)
        try:
            return next(self._it[key])
        except StopIteration:
            if key in self._sp:
                self._sp[key].shuffle()
            self._it[key] = self._make(key)
            return next(self._it[key])
