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
# lapply(seq_along(xs), function(k) fields::Tps(xs[[k]], Ys[[k]])) in ONE library call: the tiles of V73:690-738 (or the
# response layers of one station table); every fit with 8..256 distinct locations is one workgroup of one kernel launch.
# Returns a list of "machisplin_tps" objects, NULL where a fit failed (the loop of V73:707-738 skips such a tile's spline).
mhs_Tps_many <- function(xs, Ys, lambda = NA_real_) {
  hs <- .Call("mhsr_tps_fit_many", lapply(xs, as.matrix), lapply(Ys, function(y) as.numeric(unlist(y))), lambda, 0L)
  lapply(hs, function(h) if (is.null(h)) NULL else structure(list(handle = h), class = "machisplin_tps"))
}
predict.machisplin_tps <- function(object, x, ...) .Call("mhsr_tps_predict_points", object$handle, as.matrix(x))
mhs_interpolate <- function(r, object) {
  v <- .Call("mhsr_tps_predict_grid", object$handle, .mhs_geom(r), c(0L, nrow(r), 0L, ncol(r)))
  terra::setValues(terra::rast(r), v)
}


# ---- learner fits on the device (SURVEY.md 8f rank 4); every one keeps the CRAN call as its fallback -----------------
# kernlab::ksvm(mod.form, data = dat) (V73:251, V73:560).  sigma: kernlab draws it with sigest() from a random half
# of the rows -- do the same here so that the two backends see the same kernel width.
.mhs_ksvm <- function(dat, sigma = mean(kernlab::sigest(as.matrix(dat[, -1]), scaled = TRUE)[c(1, 3)])) {
  X <- as.matrix(dat[, -1]); p <- ncol(X)
  f <- .Call("mhsr_svr_fit", X, as.numeric(dat[, 1]), sigma, 1.0, 0.1, 0.001)
  sv <- which(f[[1]] != 0)
  Z <- sweep(sweep(X[sv, , drop = FALSE], 2, f[[3]]), 2, f[[4]], "/")
  list(handle = .Call("mhsr_svr_load", as.numeric(f[[1]][sv]), as.numeric(t(Z)), p, f[[2]], sigma, f[[3]], f[[4]], f[[5]], f[[6]]),
       beta = f[[1]], b = f[[2]], sigma = sigma, iterations = f[[7]])
}
# nnet::nnet(mod.form, data = trainNN, size = 10, linout = TRUE, maxit = 10000) (V73:249, V73:463); trainNN$resp is
# already (resp - min) / max (V73:455-459)
.mhs_nnet <- function(trainNN, max2.resp.f, min.resp.f, maxit = 10000L) {
  X <- as.matrix(trainNN[, -1]); p <- ncol(X)
  f <- .Call("mhsr_nnet_fit", X, as.numeric(trainNN[, 1]), runif((p + 1) * 10 + 11, -0.7, 0.7), as.integer(maxit))
  list(handle = .Call("mhsr_nnet_load", f[[1]], p, 10L, max2.resp.f, min.resp.f), wts = f[[1]], value = f[[2]], convergence = f[[4]])
}
# inside machisplin.gbm.step's loop (V73:1843, 1919): the hold-out predictions of fold model i for every stage at once,
# instead of one predict.gbm per gbm.more
.mhs_gbm_holdout_stages <- function(handle, x.holdout, step.size, n.fitted)
  .Call("mhsr_gbm_staged_points", handle, as.matrix(x.holdout), as.integer(step.size), as.integer(n.fitted))


# ---- Step 2 as it is threaded through machisplin.mltps (V73:442-619), all six members -----------------------------------
# The loop  for(k in mods.run)  keeps its fits, its variable-importance lines and its station residuals (O(stations) work in
# the packages); only the raster side of each  if (k=="x")  block -- the terra::predict(rast_stack, model) pair and the
# pred.elev accumulation -- is replaced by ONE call that records the fitted member, and the division by the weight total
# after the loop by ONE call that evaluates all recorded members in a single pass over the raster.  The edit list, by V73
# line (INTEGRATION.md section 2 repeats it):
#   before V73:447   hip <- .mhs_ready(); S2 <- .mhs_step2_new()
#   V73:468-475 (n)  if (hip) S2 <- .mhs_step2_member(S2, "n", mod.nn.tps.FINAL,   OptX.mfit.wt[[iter.mod]], n.covars, max2.resp.f, min.resp.f) else { <the eight lines> }
#   V73:497-499 (b)  if (hip) S2 <- .mhs_step2_member(S2, "b", mod.brt.tps.FINAL,  OptX.mfit.wt[[iter.mod]], n.covars) else { <the three lines> }
#   V73:521-523 (r)  if (hip) S2 <- .mhs_step2_member(S2, "r", mod.rf.tps.FINAL,   OptX.mfit.wt[[iter.mod]], n.covars) else { ... }
#   V73:543-545 (m)  if (hip) S2 <- .mhs_step2_member(S2, "m", mod.MARS.tps.FINAL, OptX.mfit.wt[[iter.mod]], n.covars) else { ... }
#   V73:582-584 (v)  if (hip) S2 <- .mhs_step2_member(S2, "v", mod.SVM.tps.FINAL,  OptX.mfit.wt[[iter.mod]], n.covars) else { ... }
#   V73:604-606 (g)  if (hip) S2 <- .mhs_step2_member(S2, "g", mod.GAM.tps.FINAL,  OptX.mfit.wt[[iter.mod]], n.covars) else { ... }
#   V73:619          pred.elev <- if (hip) .mhs_step2_finish(S2, covar.ras, OptX.mfit.wt.tot) else (pred.elev/OptX.mfit.wt.tot)
# The members are summed in the loop's order (mods.run), each times its weight, then divided by the unrounded total -- the
# arithmetic of V73:471-475 ... 606 and 619.
.mhs_step2_new <- function() list(handles = list(), wts = numeric(0))
.mhs_step2_member <- function(S2, k, mod, wt, n.covars, max2.resp.f = 1, min.resp.f = 0) {
  S2$handles[[length(S2$handles) + 1L]] <- .mhs_model(k, mod, n.covars, max2.resp.f, min.resp.f)
  S2$wts <- c(S2$wts, as.numeric(wt))
  S2
}
.mhs_step2_finish <- function(S2, covar.ras, OptX.mfit.wt.tot) mhs_ensemble_raster(covar.ras, S2$handles, S2$wts, OptX.mfit.wt.tot)
