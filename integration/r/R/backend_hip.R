# R side of the drop-in (UNTESTED HERE: no R in the build image).  Each helper replaces one
# call site of machisplin.mltps (V73 = R/ensemble.machine.learning.thin.plate.splines.V73.R)
# and returns the same kind of object the caller consumes (a SpatRaster), so the 7 exported
# functions and their return structure are unchanged.  Selected with
#   options(machisplin.backend = "hip")
# and falling back to the original CRAN call when the library cannot be initialised.

.mhs_geom <- function(r) c(terra::xmin(r), terra::ymax(r), terra::res(r)[1], terra::res(r)[2], nrow(r), ncol(r))

.mhs_ready <- function() {
  if (!identical(getOption("machisplin.backend", "cpu"), "hip")) return(FALSE)
  isTRUE(tryCatch({ .Call("mhsr_init", 0L); TRUE }, error = function(e) FALSE))
}

# ---- flat parameter arrays of the fitted members (loaders in include/machisplin_hip.h) ----
.mhs_model <- function(k, mod, n.covars, max2.resp.f = 1, min.resp.f = 0) {
  p <- n.covars
  switch(k,
    g = .Call("mhsr_lm_load", as.numeric(mod$coefficients)),
    n = .Call("mhsr_nnet_load", as.numeric(mod$wts), p, as.integer(mod$n[2]), max2.resp.f, min.resp.f),
    m = { sel <- mod$selected.terms
          .Call("mhsr_earth_load", as.numeric(mod$coefficients), as.integer(t(mod$dirs[sel, , drop = FALSE])),
                as.numeric(t(mod$cuts[sel, , drop = FALSE])), p) },
    v = { sc <- kernlab::scaling(mod)
          .Call("mhsr_svr_load", as.numeric(unlist(kernlab::coef(mod))), as.numeric(t(kernlab::xmatrix(mod))), p,
                kernlab::b(mod), kernlab::kpar(kernlab::kernelf(mod))$sigma,
                sc$x.scale$`scaled:center`, sc$x.scale$`scaled:scale`,
                sc$y.scale$`scaled:center`, sc$y.scale$`scaled:scale`) },
    b = { nt <- mod$gbm.call$best.trees; tr <- mod$trees[seq_len(nt)]
          cnt <- vapply(tr, function(t) length(t[[1]]), 1L)
          .Call("mhsr_gbm_load", mod$initF, as.numeric(c(0, cumsum(cnt))),
                as.integer(unlist(lapply(tr, `[[`, 1))), as.numeric(unlist(lapply(tr, `[[`, 2))),
                as.integer(unlist(lapply(tr, `[[`, 3))), as.integer(unlist(lapply(tr, `[[`, 4))),
                as.integer(unlist(lapply(tr, `[[`, 5))), p) },
    r = { f <- mod$forest; nb <- f$ndbigtree
          pick <- function(m) unlist(lapply(seq_len(f$ntree), function(t) m[seq_len(nb[t]), t]))
          .Call("mhsr_rf_load", as.numeric(c(0, cumsum(nb))), as.integer(pick(f$leftDaughter)),
                as.integer(pick(f$rightDaughter)), as.integer(pick(f$nodestatus)), as.integer(pick(f$bestvar)),
                as.numeric(pick(f$xbestsplit)), as.numeric(pick(f$nodepred)), p) })
}

# ---- Step 2: pred.elev over the whole raster (replaces the six terra::predict calls and the
#      weighted accumulation, V73:447-619) --------------------------------------------------
mhs_ensemble_raster <- function(covar.ras, handles, OptX.mfit.wt, OptX.mfit.wt.tot) {
  v <- .Call("mhsr_ensemble_predict", handles, as.numeric(OptX.mfit.wt), OptX.mfit.wt.tot,
             .mhs_geom(covar.ras), terra::values(covar.ras))
  terra::setValues(covar.ras[[1]], v)
}

# ---- Step 3 + 4: final.TPS (replaces V73:636-897) ------------------------------------------
mhs_tps_surface <- function(rast_stack, dat, res.FINAL, n.covars, tile.edge = 1500L, lambda = NA_real_) {
  xy <- as.matrix(dat[, c(n.covars, n.covars + 1)])          # LONG, LAT columns (V73:688,751)
  v <- .Call("mhsr_tps_surface", .mhs_geom(rast_stack), xy, as.numeric(res.FINAL), as.numeric(dat[, 2]),
             as.integer(tile.edge), lambda, 0L)
  terra::setValues(terra::rast(rast_stack[[1]]), v)
}

# How terra::interpolate's replacement sums the knots: "auto" (by cost), "direct" (predict.Krig's own loop, results
# bit-identical across windows -- use it when comparing against captured fields output) or "far.field".
mhs_tps_eval_mode <- function(mode = c("auto", "direct", "far.field")) {
  mode <- match.arg(mode)
  invisible(.Call("mhsr_tps_eval_mode", match(mode, c("auto", "direct", "far.field")) - 1L))
}


# ---- a drop-in for the fields::Tps object itself (keeps the R loop of V73:690-753 as it is) ----
# mhs_Tps(xy, y) returns an object of class "machisplin_tps"; terra::interpolate(rast, obj) then works through
# the predict method below (points, block by block), and mhs_interpolate() is the whole-grid fast path.
mhs_Tps <- function(x, Y, lambda = NA_real_) {
  structure(list(handle = .Call("mhsr_tps_fit", as.matrix(x), as.numeric(unlist(Y)), lambda, 0L)),
            class = "machisplin_tps")
}
predict.machisplin_tps <- function(object, x, ...) .Call("mhsr_tps_predict_points", object$handle, as.matrix(x))
mhs_interpolate <- function(r, object) {
  v <- .Call("mhsr_tps_predict_grid", object$handle, .mhs_geom(r), c(0L, nrow(r), 0L, ncol(r)))
  terra::setValues(terra::rast(r), v)
}
