graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  node [
    id 9
    label "9"
  ]
  node [
    id 10
    label "10"
  ]
  node [
    id 11
    label "11"
  ]
  node [
    id 12
    label "12"
  ]
  node [
    id 13
    label "13"
  ]
  edge [
    source 0
    target 1
    capacity 4510
  ]
  edge [
    source 0
    target 2
    capacity 2580
  ]
  edge [
    source 0
    target 7
    capacity 2328
  ]
  edge [
    source 1
    target 2
    capacity 2299
  ]
  edge [
    source 1
    target 3
    capacity 3339
  ]
  edge [
    source 1
    target 0
    capacity 4510
  ]
  edge [
    source 2
    target 5
    capacity 2874
  ]
  edge [
    source 2
    target 0
    capacity 2580
  ]
  edge [
    source 2
    target 1
    capacity 2299
  ]
  edge [
    source 3
    target 4
    capacity 3809
  ]
  edge [
    source 3
    target 10
    capacity 1882
  ]
  edge [
    source 3
    target 1
    capacity 3339
  ]
  edge [
    source 4
    target 5
    capacity 2674
  ]
  edge [
    source 4
    target 6
    capacity 4041
  ]
  edge [
    source 4
    target 3
    capacity 3809
  ]
  edge [
    source 5
    target 9
    capacity 1536
  ]
  edge [
    source 5
    target 12
    capacity 2601
  ]
  edge [
    source 5
    target 2
    capacity 2874
  ]
  edge [
    source 5
    target 4
    capacity 2674
  ]
  edge [
    source 6
    target 7
    capacity 2959
  ]
  edge [
    source 6
    target 4
    capacity 4041
  ]
  edge [
    source 7
    target 8
    capacity 4853
  ]
  edge [
    source 7
    target 0
    capacity 2328
  ]
  edge [
    source 7
    target 6
    capacity 2959
  ]
  edge [
    source 8
    target 9
    capacity 2604
  ]
  edge [
    source 8
    target 11
    capacity 2690
  ]
  edge [
    source 8
    target 13
    capacity 3298
  ]
  edge [
    source 8
    target 7
    capacity 4853
  ]
  edge [
    source 9
    target 5
    capacity 1536
  ]
  edge [
    source 9
    target 8
    capacity 2604
  ]
  edge [
    source 10
    target 11
    capacity 2068
  ]
  edge [
    source 10
    target 13
    capacity 2345
  ]
  edge [
    source 10
    target 3
    capacity 1882
  ]
  edge [
    source 11
    target 12
    capacity 4133
  ]
  edge [
    source 11
    target 8
    capacity 2690
  ]
  edge [
    source 11
    target 10
    capacity 2068
  ]
  edge [
    source 12
    target 13
    capacity 4375
  ]
  edge [
    source 12
    target 5
    capacity 2601
  ]
  edge [
    source 12
    target 11
    capacity 4133
  ]
  edge [
    source 13
    target 8
    capacity 3298
  ]
  edge [
    source 13
    target 10
    capacity 2345
  ]
  edge [
    source 13
    target 12
    capacity 4375
  ]
]
