# R side of the multi-GPU drop-in (UNTESTED IN R: no R in the build image; the .Call entry points below are executed against
# a stub R runtime in tests/test_r_shim_exec.py, and the library calls behind them in tests/test_multi_gpu.py).
#
# ONE R process drives every GPU of the node: mhs_init_devices(8) brings up eight device slots inside libmachisplin_hip.so,
# and each of the two calls below hands the WHOLE job to the library, which runs one host thread and one set of HIP streams
# per GPU and moves planes between GPUs itself (RCCL all-gather / peer copies over xGMI).  Nothing travels through R's
# serialisation and there are no PSOCK workers (the round-4 design, priced by its own comment at minutes of R time).
#
# The reference's parallelism: snowfall's sfLapply over response layers in old versions (old/...V69.R:964), none today
# (n.cores forced to 1, V73:117); its README runs machisplin.mltps once per tile of machisplin.tiles.create and merges with
# machisplin.tiles.merge (README.md:157-215).  Model FITTING (Step 1 and the final fits, V73:176-436) stays in R and in the
# CRAN packages, single-threaded as the reference runs it; what goes to the GPUs is everything raster-sized.

mhs_init_devices <- function(n.gpus, ids = NULL) {
  options(machisplin.n.slots = as.integer(n.gpus), machisplin.slot0.share = NA_real_)
  .Call("mhsr_init_devices", as.integer(n.gpus), if (is.null(ids)) NULL else as.integer(ids))
}

# The two calls below keep their device buffers between calls (a layer loop reuses them); a long session gets the memory back with
mhs_multi_trim <- function() invisible(.Call("mhsr_multi_trim"))

# ---- machisplin.mltps Steps 2-5 of ONE response layer over all GPUs (replaces V73:447-930 behind `if (hip)`) -------------
# handles / OptX.mfit.wt / OptX.mfit.wt.tot: the fitted members as .mhs_step2_member collects them (mods.run order, rounded
# kept weights, unrounded total -- V73:337-392); dat: dat_tps[[i]] (resp, covariates, LONG, LAT -- V73:145-154).
# Returns list(final = SpatRaster, rsq.model, rsq.final, lambda, used.tps): `final` is what V73:925-930 selects.
mhs_mltps_multi <- function(covar.ras, handles, OptX.mfit.wt, OptX.mfit.wt.tot, dat, tile.edge = 1500L, lambda = NA_real_,
                            slot0.share = getOption("machisplin.slot0.share", NA_real_)) {
  X <- as.matrix(dat[, -1, drop = FALSE])
  out <- .Call("mhsr_mltps_grid_multi", handles, as.numeric(OptX.mfit.wt), OptX.mfit.wt.tot, .mhs_geom(covar.ras),
               terra::values(covar.ras), X, as.numeric(dat[, 1]), as.integer(tile.edge), lambda, 0L, slot0.share)
  # what this call measured balances the next layer's bands -- but only a change of more than 5 % of a band is taken over:
  # the measurement jitters, and a band plan that moves by one 16-row unit makes the library rebuild its cached device buffers
  n.slots <- max(1L, as.integer(getOption("machisplin.n.slots", 1L)))
  if (is.na(slot0.share) || is.na(out[[7]]) || abs(out[[7]] - slot0.share) > 0.05 / n.slots) options(machisplin.slot0.share = out[[7]])
  list(final = terra::setValues(covar.ras[[1]], out[[1]]), rsq.model = out[[2]], rsq.final = out[[3]], lambda = out[[4]],
       used.tps = out[[5]] == 1L, n.slots = out[[6]])
}

# ---- a machisplin.tiles.* run over all GPUs (README.md:157-215) ------------------------------------------------------------
#   tiles <- machisplin.tiles.create(int.values, covar.ras, out.ncol = 2, out.nrow = 2, feather.d = 50)
#   fits  <- mhs_tiles_fit(tiles, ...)                     # Step 1 + the final fits per (tile, layer): in R, unchanged
#   bio   <- mhs_tiles_mltps(covar.ras, tiles, fits, out.ncol = 2, out.nrow = 2, feather.d = 50)   # list of merged SpatRasters
#
# fits[[l]][[t]] = list(handles, wts, wt.tot, dat) for response layer l and tile t (tiles row-major from the south-west,
# V73:1192-1197): the members fitted on tile t's stations, their weights, and the tile's dat_tps (resp, covariates, LONG, LAT
# at the TILE raster's cell centres).  The library crops covar.ras to every tile itself (mhs_tiles_create_windows repeats
# V73:1165-1208), runs unit (l, t) on GPU ((l - 1) * n.tiles + t - 1) %% n.gpus, brings a layer's tiles to its owner over xGMI
# and merges them as machisplin.tiles.merge does (V73:1392-1548).
mhs_tiles_mltps <- function(covar.ras, tiles, fits, out.ncol, out.nrow, feather.d = 50, tps = TRUE, tile.edge = 1500L,
                            lambda = NA_real_) {
  n.layers <- length(fits)
  n.tiles <- out.ncol * out.nrow
  units <- vector("list", n.layers * n.tiles)
  for (l in seq_len(n.layers)) for (t in seq_len(n.tiles)) {
    f <- fits[[l]][[t]]
    units[[(l - 1L) * n.tiles + t]] <- list(f$handles, as.numeric(f$wts), f$wt.tot, as.matrix(f$dat[, -1, drop = FALSE]),
                                            as.numeric(f$dat[, 1]))
  }
  out <- .Call("mhsr_tiles_units_multi", .mhs_geom(covar.ras), terra::values(covar.ras), as.integer(out.ncol),
               as.integer(out.nrow), feather.d, units, as.integer(n.layers), as.integer(tps), as.integer(tile.edge), lambda, 0L)
  list(final = lapply(out[[1]], function(v) terra::setValues(covar.ras[[1]], v)),
       rsq = array(out[[2]], c(2L, n.tiles, n.layers), dimnames = list(c("rsq.model", "rsq.final"), NULL, NULL)))
}
